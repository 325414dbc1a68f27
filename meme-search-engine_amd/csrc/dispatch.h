// Cross-thread query coalescer: the meeting point of the reference's call shape.
//
// The reference answers every request with ONE query from its own thread: the small-scale server holds a shared read guard
// and calls `index.search(&query, k)` per request (src/main.rs:896-934,1043-1049); the disk-index server runs a thread per
// core, each with its own Scratch, one search per request (src/query_disk_index.rs:711-736).  On a CPU that is the right
// shape -- a core per query.  On one MI355X a pass over the rows costs the same for 1 query as for 128, so T callers
// arriving with one query each must share a pass: callers enqueue and block, ONE worker thread per handle gathers what is
// waiting and runs a single batched pass, then hands every caller exactly its own rows.
//
// Gather rule (no fixed batching delay for a lone caller): the worker fires when its target is waiting -- the queries that
// queued up during the last pass plus the ones that pass answered (closed-loop callers are back within microseconds) -- or
// `max_queries`, or when the oldest waiting request is `max_wait` old and the callers just answered had a grace period
// (<= 1 ms) to return, whichever comes first.  A single caller therefore never waits (target 1), T closed-loop callers settle
// at T queries per pass after two passes, and callers that do not come back cost a grace period on every other pass at most.
// More expected callers than one pass holds are served in EQUAL passes (512 callers, 320 per pass: 256 + 256, not 320 + 192).
//
// Round 5: a handle may run SEVERAL workers (the graph's request path runs two: one pass's upload / download and host side
// overlap the other's kernels).  One worker gathers at a time, by the same rule; the others run or sleep.  Completion is
// signalled through a ring of (mutex, condition variable) slots that requests are assigned to in arrival order, 256 to a slot:
// a pass wakes the callers of the slots it touched instead of everybody who is waiting (4096 callers, 1024 per pass: one
// shared variable woke 3072 sleepers for nothing on every pass and sent all of them through one mutex).
#pragma once
#include "common.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mse {

struct DispatchReq {
    // what the caller asked for (meaning is the owner's: f16 or f32 queries, i64 or f32 scores ...)
    const void* queries = nullptr;
    size_t nq = 0, k = 0;
    void* out_a = nullptr;
    void* out_b = nullptr;
    const void* aux0 = nullptr;       // owner-defined grouping keys / extra inputs
    const void* aux1 = nullptr;
    const void* aux2 = nullptr;
    size_t aux_n = 0;
    // filled by the worker
    int rc = 0;
    std::string err;
    uint32_t flags = 0;
    // queue plumbing
    bool done = false;
    uint32_t slot = 0;                // completion slot (Coalescer::wake_), assigned on arrival
    std::chrono::steady_clock::time_point t_arrive;
};

struct DispatchStats {
    uint64_t queries = 0, requests = 0, passes = 0, max_pass_queries = 0, deadline_fires = 0, retried_alone = 0;
};

class Coalescer {
  public:
    // run(batch): executes the requests of one pass and sets rc / err of each.  Called on the worker thread only, after
    // `on_thread_start` ran there once (device selection).
    using RunFn = std::function<void(std::vector<DispatchReq*>&)>;
    Coalescer(size_t max_queries, uint32_t max_wait_us, RunFn run, std::function<void()> on_thread_start, int n_workers = 1);
    ~Coalescer();   // requests still queued are answered with an error; the workers are joined
    // index of the calling worker thread within its Coalescer (0 .. n_workers-1; 0 on any other thread): run() uses it to pick
    // its own scratch when a handle has several workers
    static int worker_index();
    int n_workers() const { return (int)workers_.size(); }
    // blocks until the request was executed; returns its rc and leaves its message in the thread's mse_last_error()
    int submit(DispatchReq& r);
    DispatchStats stats();
    size_t max_queries() const { return max_queries_; }
    size_t target() const;   // queries the next pass waits for (call with mu_ held)
    uint32_t max_wait_us() const { return max_wait_us_.load(); }
    void set_max_wait_us(uint32_t us) { max_wait_us_.store(us); }   // takes effect from the next gather

  private:
    void loop(int index);
    static constexpr uint32_t WAKE_SLOTS = 64, WAKE_RUN = 256;   // 256 consecutive arrivals share a slot
    struct WakeSlot { std::mutex mu; std::condition_variable cv; };
    const size_t max_queries_;
    std::atomic<uint32_t> max_wait_us_;
    RunFn run_;
    std::function<void()> on_start_;
    std::mutex mu_;
    std::condition_variable cv_worker_;
    // Completion: a variable per request would be hundreds of futex wake-ups per pass; ONE variable for all callers wakes every
    // sleeper of every other pass too.  A ring of slots, 256 consecutive arrivals to a slot (the queue is first in, first out, so a
    // pass touches few slots): `done` is set under the slot's mutex, the slot's variable is signalled after the unlock.
    WakeSlot wake_[WAKE_SLOTS];
    uint64_t arrivals_ = 0;
    std::deque<DispatchReq*> queue_;
    size_t queued_queries_ = 0;
    size_t expect_ = 1;
    bool stop_ = false;
    bool gathering_ = false;          // one worker gathers at a time
    bool expected_returners_ = false;
    std::chrono::steady_clock::time_point grace_until_ = std::chrono::steady_clock::time_point::min();
    DispatchStats st_;
    std::vector<std::thread> workers_;
};

// Writer-preferring shared/exclusive lock: searches share, `add` excludes -- the RwLock of src/main.rs:1016 (write) and
// :1046 (read).  (pthread's default rwlock prefers readers: under a steady stream of searches an `add` would never get in.)
class SharedExclusive {
  public:
    void lock_shared() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !writer_ && writers_waiting_ == 0; });
        readers_++;
    }
    void unlock_shared() {
        std::unique_lock<std::mutex> lk(mu_);
        if (--readers_ == 0) cv_.notify_all();
    }
    void lock() {
        std::unique_lock<std::mutex> lk(mu_);
        writers_waiting_++;
        cv_.wait(lk, [&] { return !writer_ && readers_ == 0; });
        writers_waiting_--;
        writer_ = true;
    }
    void unlock() {
        std::unique_lock<std::mutex> lk(mu_);
        writer_ = false;
        cv_.notify_all();
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    size_t readers_ = 0, writers_waiting_ = 0;
    bool writer_ = false;
};

// wait budget when the caller gives none: a tenth of one pass over the rows at ~4 TB/s, between 200 us and 5 ms
uint32_t default_wait_us(size_t n_rows, size_t row_bytes);

}  // namespace mse
