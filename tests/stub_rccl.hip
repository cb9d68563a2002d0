// TEST INFRASTRUCTURE (never loaded by the product): a stand-in for librccl.so that lets SEVERAL ranks of the native obstacle
// exchange (include/rmp2.h rmp2_exchange_*) live in ONE process on ONE GPU, so that the nranks > 1 branch of the exchange --
// rank-offset slices, system-scope `ready` events, the GPU-side wait that world 1 drops -- runs on the one-GPU test box.
// It exports the five RCCL entry points the exchange binds at run time (csrc/rmp2_hip.hip load_rccl).  The all-gather is
// stream-ordered device-to-device copies: rank r's stream waits for every peer's "my slice is ready" event of the same
// call number, copies every slice into its own table, and signals "I have read your slice"; a rank's call completes on its
// stream when its table is filled and all peers have read its slice -- the contract of ncclAllGather.  Ranks are driven
// from one host thread each (the host side rendez-vous blocks, with a time-out that turns a mismatch into an error
// instead of a hang).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr int kTimeoutS = 20;
struct Uid { char bytes[128]; };

struct Call {
  const void* send[kMaxRanks] = {};
  hipEvent_t sent[kMaxRanks] = {};
  hipEvent_t copied[kMaxRanks] = {};
  int arrived = 0, done = 0;
};

struct Group {
  int nranks = 0, joined = 0, alive = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<uint64_t, Call> calls;
  uint64_t seq[kMaxRanks] = {};
  std::vector<hipEvent_t> events;   // destroyed with the group (an event may still be awaited when its call returns)
};
struct Comm { Group* g; int rank; };

std::mutex g_mutex;
std::map<std::string, Group*> g_groups;
uint64_t g_uid_counter = 0;
uint64_t g_allgathers = 0;

template <class Pred> bool wait_for(Group* g, std::unique_lock<std::mutex>& lk, Pred p) {
  return g->cv.wait_for(lk, std::chrono::seconds(kTimeoutS), p);
}
}  // namespace

extern "C" {

int ncclGetUniqueId(Uid* uid) {
  std::lock_guard<std::mutex> lk(g_mutex);
  std::memset(uid->bytes, 0, sizeof(uid->bytes));
  const uint64_t id = ++g_uid_counter;
  std::memcpy(uid->bytes, "stub-rccl", 9);
  std::memcpy(uid->bytes + 16, &id, sizeof(id));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, Uid uid, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
  Group* g;
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    Group*& slot = g_groups[std::string(uid.bytes, sizeof(uid.bytes))];
    if (!slot) slot = new Group(), slot->nranks = nranks;
    g = slot;
  }
  std::unique_lock<std::mutex> lk(g->m);
  if (g->nranks != nranks) return 4;
  ++g->joined, ++g->alive;
  g->cv.notify_all();
  if (!wait_for(g, lk, [&] { return g->joined >= g->nranks; })) return 2;  // ncclSystemError: a peer never joined
  *comm = new Comm{g, rank};
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || !send || !recv || dtype != 7) return 4;  // ncclFloat32 only
  Group* g = c->g;
  const int r = c->rank, n = g->nranks;
  const size_t bytes = count * sizeof(float);
  hipEvent_t sent = nullptr, copied = nullptr;
  if (hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess) return 1;
  if (hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess) return 1;
  if (hipEventRecord(sent, stream) != hipSuccess) return 1;   // "my slice is ready": everything before this call on my stream
  Call* call;
  {
    std::unique_lock<std::mutex> lk(g->m);
    const uint64_t k = g->seq[r]++;
    call = &g->calls[k];
    call->send[r] = send, call->sent[r] = sent, call->copied[r] = copied;
    g->events.push_back(sent), g->events.push_back(copied);
    ++call->arrived;
    g->cv.notify_all();
    if (!wait_for(g, lk, [&] { return call->arrived >= n; })) return 2;
  }
  for (int j = 0; j < n; ++j) {
    if (j != r && hipStreamWaitEvent(stream, call->sent[j], 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(static_cast<char*>(recv) + j * bytes, call->send[j], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  }
  if (hipEventRecord(copied, stream) != hipSuccess) return 1;  // "I have read every slice"
  {
    std::unique_lock<std::mutex> lk(g->m);
    ++call->done;
    g->cv.notify_all();
    if (!wait_for(g, lk, [&] { return call->done >= n; })) return 2;
  }
  for (int j = 0; j < n; ++j)   // my slice may be rewritten by later work on my stream only once every peer has read it
    if (j != r && hipStreamWaitEvent(stream, call->copied[j], 0) != hipSuccess) return 1;
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    ++g_allgathers;
  }
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  Group* g = c->g;
  bool last;
  {
    std::lock_guard<std::mutex> lk(g->m);
    last = --g->alive == 0;
  }
  delete c;
  if (last) {
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : g->events) (void)hipEventDestroy(e);
    std::lock_guard<std::mutex> lk(g_mutex);
    for (auto it = g_groups.begin(); it != g_groups.end(); ++it)
      if (it->second == g) { g_groups.erase(it); break; }
    delete g;
  }
  return 0;
}

int ncclCommCount(void* comm, int* count) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || !count) return 4;
  // test hook: STUB_RCCL_LIE_ABOUT_COUNT=1 makes the communicator report one rank more than it has (a communicator that formed
  // with another size than the caller asked for must be refused by rmp2_exchange_create)
  const char* lie = std::getenv("STUB_RCCL_LIE_ABOUT_COUNT");
  *count = c->g->nranks + ((lie && lie[0] == '1') ? 1 : 0);
  return 0;
}

int ncclCommUserRank(void* comm, int* rank) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || !rank) return 4;
  *rank = c->rank;
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "stub-rccl: success";
    case 1: return "stub-rccl: HIP call failed";
    case 2: return "stub-rccl: a peer rank did not arrive within the time-out";
    case 4: return "stub-rccl: invalid argument";
    default: return "stub-rccl: error";
  }
}

// test hook: all-gather calls completed (summed over ranks) since the library was loaded
uint64_t stub_rccl_allgathers(void) {
  std::lock_guard<std::mutex> lk(g_mutex);
  return g_allgathers;
}

}  // extern "C"
