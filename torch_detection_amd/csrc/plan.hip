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
#include <mutex>
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
};
std::mutex g_mu;
Plan* g_rec = nullptr;      // plan being recorded (process-wide: autograd may issue launches from its own thread)
}  // namespace

bool tdn_plan_recording() { return g_rec != nullptr; }

void tdn_plan_push(hipStream_t stream, std::function<void(hipStream_t)> fn) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rec) g_rec->cmds.push_back(Cmd{0, stream, -1, std::move(fn)});
}

extern "C" int tdn_plan_begin(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  TDN_CHECK(g_rec == nullptr, "tdn_plan_begin: a plan is already being recorded");
  g_rec = new Plan();
  return 0;
}

extern "C" void* tdn_plan_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* p = g_rec;
  g_rec = nullptr;
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
  if (!g_rec) return -1;
  const int id = g_rec->nevents++;
  g_rec->cmds.push_back(Cmd{1, (hipStream_t)stream, id, nullptr});
  return id;
}

// "`stream` was made to wait for event `event_id` here"
extern "C" int tdn_plan_stream_wait(void* stream, int event_id) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_rec) return 0;
  TDN_CHECK(event_id >= 0 && event_id < g_rec->nevents, "tdn_plan_stream_wait: unknown event %d", event_id);
  g_rec->cmds.push_back(Cmd{2, (hipStream_t)stream, event_id, nullptr});
  return 0;
}

extern "C" int tdn_plan_run(void* plan) {
  TDN_CHECK(plan != nullptr, "tdn_plan_run: NULL plan");
  TDN_CHECK(!tdn_plan_recording(), "tdn_plan_run: cannot run a plan while another one is being recorded");
  Plan* p = (Plan*)plan;
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
  return 0;
}

extern "C" int tdn_plan_free(void* plan) {
  if (!plan) return 0;
  Plan* p = (Plan*)plan;
  for (hipEvent_t e : p->events)
    if (e) (void)hipEventDestroy(e);
  delete p;
  return 0;
}
