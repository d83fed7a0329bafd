// Host-side stream-hazard check of the engine's multi-stream launch sequence (mp_model_config::debug bit 0; include/manipose_hip.h,
// "stream-hazard check").  No device code: the engine declares, for every launch, the stream it goes to and the byte ranges it reads and writes,
// and every hipEventRecord / hipStreamWaitEvent it issues; the tracker keeps one vector clock per stream (a launch ticks its stream's own
// component, an event carries the recording stream's clock, a wait takes the component-wise maximum) and reports every pair of launches on
// DIFFERENT streams that touch overlapping bytes, at least one of them writing, without a happens-before path between them - a missing
// event.  What it cannot see: what a kernel really touches (the declarations are the engine's own account of its kernels' operands), and
// host-side synchronisation (none is used inside the engine).  Pure C++: the same class is driven directly through the mp_hazard_* entry
// points by a CPU test (tests/test_host_cpu.py).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace mp {

struct HzAccess {
  const void* p;
  size_t bytes;
  bool write;
};
inline HzAccess hz_r(const void* p, double bytes) { return HzAccess{p, (size_t)bytes, false}; }
inline HzAccess hz_w(const void* p, double bytes) { return HzAccess{p, (size_t)bytes, true}; }

class HazardTracker {
 public:
  static constexpr int MAXS = 16;
  struct Clock { uint64_t c[MAXS] = {}; };

  int stream_id(const void* stream) {                 // small integers in order of first appearance
    for (size_t i = 0; i < streams_.size(); ++i)
      if (streams_[i] == stream) return (int)i;
    if ((int)streams_.size() == MAXS) {
      // more distinct streams than clocks: two streams would share an id and `r.stream == s` would hide their conflicts - that is a
      // failed check, not a clean one
      ++n_violations_;
      if (msgs_.size() < 64) msgs_.push_back("hazard tracker: more than 16 distinct streams seen; streams share an id from here on and the check is void");
      return MAXS - 1;
    }
    streams_.push_back(stream);
    return (int)streams_.size() - 1;
  }
  void record(const void* event, int s) { events_[event] = clock_[s]; ++n_events_; }
  void wait(int s, const void* event) {
    auto it = events_.find(event);
    if (it == events_.end()) return;                  // never recorded: hipStreamWaitEvent is a no-op then
    for (int i = 0; i < MAXS; ++i) clock_[s].c[i] = std::max(clock_[s].c[i], it->second.c[i]);
  }
  // a launch on stream s with its declared accesses
  void launch(int s, const char* name, const HzAccess* acc, int n) {
    const uint64_t ts = ++clock_[s].c[s];
    ++n_launches_;
    const int id = intern(name ? name : "?");
    for (int a = 0; a < n; ++a) {
      if (acc[a].p == nullptr || acc[a].bytes == 0) continue;
      const uintptr_t lo = (uintptr_t)acc[a].p, hi = lo + acc[a].bytes;
      for (const Rec& r : recs_) {
        if (r.hi <= lo || hi <= r.lo) continue;       // no overlap
        if (!r.write && !acc[a].write) continue;      // read after read
        if (r.stream == s) continue;                  // stream order
        if (clock_[s].c[r.stream] >= r.ts) { ++n_ordered_; continue; }
        ++n_violations_;
        if (msgs_.size() < 64) {
          char buf[512];
          snprintf(buf, sizeof(buf), "%s: '%s' (stream %d) %s bytes [%#zx, %#zx) that '%s' (stream %d, its launch #%llu) %s, with no event path between them",
                   r.write ? (acc[a].write ? "WAW" : "RAW") : "WAR", names_[id].c_str(), s, acc[a].write ? "writes" : "reads",
                   (size_t)std::max(lo, r.lo), (size_t)std::min(hi, r.hi), names_[r.name].c_str(), r.stream, (unsigned long long)r.ts,
                   r.write ? "wrote" : "read");
          msgs_.push_back(buf);
        }
      }
    }
    for (int a = 0; a < n; ++a)
      if (acc[a].p != nullptr && acc[a].bytes != 0)
        recs_.push_back(Rec{(uintptr_t)acc[a].p, (uintptr_t)acc[a].p + acc[a].bytes, acc[a].write, s, ts, id});
  }
  // forget the accesses every stream has synchronised past (called between steps: the history stays bounded, nothing unordered is dropped)
  void prune() {
    Clock mn;
    for (int i = 0; i < MAXS; ++i) {
      uint64_t v = UINT64_MAX;
      for (size_t s = 0; s < streams_.size(); ++s) v = std::min(v, clock_[s].c[i]);
      mn.c[i] = streams_.empty() ? 0 : v;
    }
    recs_.erase(std::remove_if(recs_.begin(), recs_.end(), [&](const Rec& r) { return r.ts <= mn.c[r.stream]; }), recs_.end());
  }
  int64_t launches() const { return n_launches_; }
  int64_t ordered_pairs() const { return n_ordered_; }
  int64_t violations() const { return n_violations_; }
  int64_t events() const { return n_events_; }
  int64_t live_records() const { return (int64_t)recs_.size(); }
  const std::vector<std::string>& messages() const { return msgs_; }

 private:
  int intern(const char* name) {                      // one copy per distinct launch name (the history of a long run stays small)
    auto it = name_ids_.find(name);
    if (it != name_ids_.end()) return it->second;
    names_.push_back(name);
    return name_ids_[name] = (int)names_.size() - 1;
  }
  struct Rec { uintptr_t lo, hi; bool write; int stream; uint64_t ts; int name; };
  std::vector<const void*> streams_;
  Clock clock_[MAXS];
  std::map<const void*, Clock> events_;
  std::vector<Rec> recs_;
  std::vector<std::string> names_;
  std::map<std::string, int> name_ids_;
  std::vector<std::string> msgs_;
  int64_t n_launches_ = 0, n_ordered_ = 0, n_violations_ = 0, n_events_ = 0;
};

}  // namespace mp
