// Prepared launch lists ("plans"): record the launches of one step once, enqueue them again with one C call.
//
// What it replaces: the ~230 host-side trips of one ResNet-50-FPN forward + backward through the operator layer
// (models/backbone/resnet.py:253-268 and models/necks/fpn.py:88-125 as re-expressed in functional.py) when the step
// cannot be captured into a hipGraph — the eager path costs ~6.7 ms of host time for a ~4.4 ms GPU step.
//
// Recording keeps, per launch, the stream it went to and a closure holding the kernel and its arguments by value
// (TDN_LAUNCH in common.h); cross-stream dependencies the host expressed with events are recorded as (record, wait)
// pairs and replayed with the plan's own events.  Raw pointers inside the arguments must stay valid and mean the same
// buffers at replay time — the host side records inside a private memory pool and keeps every tensor of the recorded
// step alive (torch_detection_amd/graph.py: PreparedStep), exactly the contract of a captured graph.
#include "common.h"
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace {
struct Cmd {
  int kind;                 // 0 launch, 1 event record, 2 stream wait
  hipStream_t stream;
  int event;                // kinds 1, 2
  std::function<void(hipStream_t)> fn;   // kind 0
};
struct Plan {
  std::vector<Cmd> cmds;
  std::vector<hipEvent_t> events;
  int nevents = 0;
  int device = -1;          // the device the recording thread had selected; tdn_plan_run selects it again
  std::thread::id owner;    // the recording thread: launches of any other thread are not part of the plan
  int foreign = 0;          // launches seen from other threads while recording (reported by tdn_plan_end)
};
std::mutex g_mu;
// plan being recorded.  Read without the mutex on every launch of the library (TDN_LAUNCH), hence atomic; everything
// behind the pointer is only touched under g_mu.
std::atomic<Plan*> g_rec{nullptr};
}  // namespace

bool tdn_plan_recording() { return g_rec.load(std::memory_order_acquire) != nullptr; }

// Only the recording thread's launches belong to the plan (PreparedStep turns autograd multithreading off for the
// recorded run, so the whole step is issued from it).  A launch from another thread — a data-loader thread staging
// the next batch, box ops of another request — ran normally and is counted, not kept: replaying it every step with
// pointers the plan does not own would be silent corruption.
void tdn_plan_push(hipStream_t stream, std::function<void(hipStream_t)> fn) {
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* rec = g_rec.load(std::memory_order_relaxed);
  if (!rec) return;
  if (rec->owner != std::this_thread::get_id()) { ++rec->foreign; return; }
  rec->cmds.push_back(Cmd{0, stream, -1, std::move(fn)});
}

extern "C" int tdn_plan_begin(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  TDN_CHECK(g_rec.load() == nullptr, "tdn_plan_begin: a plan is already being recorded");
  Plan* p = new Plan();
  p->owner = std::this_thread::get_id();
  if (hipGetDevice(&p->device) != hipSuccess) p->device = -1;
  g_rec.store(p, std::memory_order_release);
  return 0;
}

extern "C" void* tdn_plan_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* p = g_rec.exchange(nullptr);
  if (!p) {
    tdn_set_error("tdn_plan_end: no plan is being recorded");
    return nullptr;
  }
  p->events.resize(p->nevents, nullptr);
  for (int i = 0; i < p->nevents; ++i) {
    if (hipEventCreateWithFlags(&p->events[i], hipEventDisableTiming) != hipSuccess) {
      tdn_set_error("tdn_plan_end: hipEventCreate failed");
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->events[j]);
      delete p;
      return nullptr;
    }
  }
  return p;
}

// "an event was recorded on `stream` here": returns the plan's id for it, or -1 when nothing is being recorded
extern "C" int tdn_plan_event_record(void* stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* rec = g_rec.load(std::memory_order_relaxed);
  if (!rec || rec->owner != std::this_thread::get_id()) return -1;
  const int id = rec->nevents++;
  rec->cmds.push_back(Cmd{1, (hipStream_t)stream, id, nullptr});
  return id;
}

// "`stream` was made to wait for event `event_id` here"
extern "C" int tdn_plan_stream_wait(void* stream, int event_id) {
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* rec = g_rec.load(std::memory_order_relaxed);
  if (!rec || rec->owner != std::this_thread::get_id()) return 0;
  TDN_CHECK(event_id >= 0 && event_id < rec->nevents, "tdn_plan_stream_wait: unknown event %d", event_id);
  rec->cmds.push_back(Cmd{2, (hipStream_t)stream, event_id, nullptr});
  return 0;
}

extern "C" int tdn_plan_run(void* plan) {
  TDN_CHECK(plan != nullptr, "tdn_plan_run: NULL plan");
  TDN_CHECK(!tdn_plan_recording(), "tdn_plan_run: cannot run a plan while another one is being recorded");
  Plan* p = (Plan*)plan;
  // the streams and pointers of the plan belong to the device it was recorded on
  int cur = -1;
  if (p->device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != p->device) {
    hipError_t e = hipSetDevice(p->device);
    TDN_CHECK(e == hipSuccess, "tdn_plan_run: hipSetDevice(%d): %s", p->device, hipGetErrorString(e));
  }
  struct Restore { int dev; ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore{cur != p->device ? cur : -1};
  for (Cmd& c : p->cmds) {
    if (c.kind == 0) {
      c.fn(c.stream);
    } else if (c.kind == 1) {
      hipError_t e = hipEventRecord(p->events[c.event], c.stream);
      TDN_CHECK(e == hipSuccess, "tdn_plan_run: hipEventRecord: %s", hipGetErrorString(e));
    } else {
      hipError_t e = hipStreamWaitEvent(c.stream, p->events[c.event], 0);
      TDN_CHECK(e == hipSuccess, "tdn_plan_run: hipStreamWaitEvent: %s", hipGetErrorString(e));
    }
  }
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_plan_stats(void* plan, int32_t* out3) {
  TDN_CHECK(plan != nullptr && out3 != nullptr, "tdn_plan_stats: NULL argument");
  Plan* p = (Plan*)plan;
  out3[0] = out3[1] = out3[2] = 0;
  for (const Cmd& c : p->cmds) out3[c.kind] += 1;
  return p->foreign;   // launches other threads made while this plan was being recorded (not part of it)
}

extern "C" int tdn_plan_free(void* plan) {
  if (!plan) return 0;
  Plan* p = (Plan*)plan;
  for (hipEvent_t e : p->events)
    if (e) (void)hipEventDestroy(e);
  delete p;
  return 0;
}
