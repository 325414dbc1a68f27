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
// Round 5: a handle may run SEVERAL workers (the graph's request path runs three: one pass's upload / download and host side
// overlap the other's kernels).  One worker gathers at a time, by the same rule; the others run or sleep.  Completion is
// signalled through a ring of futex words that requests are assigned to in arrival order, 256 to a word: a request's `done` flag
// is an atomic of its own, a pass bumps the words of the slots it touched and wakes their sleepers with ONE futex call each --
// no mutex on the way out (4096 callers, 1024 per pass: one shared condition variable woke 3072 sleepers for nothing on every
// pass and sent all 4096 through one mutex; a variable per slot still queued 256 woken callers on the slot's mutex).
#pragma once
#include "common.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
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
    std::atomic<uint32_t> done{0};    // set by the worker (release) as the LAST access to this record: it lives on its caller's stack
    uint32_t slot = 0;                // completion slot (Coalescer::wake_), assigned on arrival
    DispatchReq* next = nullptr;      // inbox link
    std::chrono::steady_clock::time_point t_arrive;
    // asynchronous requests (submit_async): nobody sleeps on them; when executed they go onto the handle's completion list, which
    // is the worker's last access to the record (its owner may free it as soon as it pops it)
    bool async = false;
    DispatchReq* cnext = nullptr;
    void* owner = nullptr;            // the object an asynchronous record is part of (the owner's to interpret)
    class CompletionQueue* cq = nullptr;   // where it is handed back (null: the handle's own queue)
};

struct DispatchStats {
    uint64_t queries = 0, requests = 0, passes = 0, max_pass_queries = 0, deadline_fires = 0, retried_alone = 0;
    uint64_t run_us = 0;              // time the workers spent inside run() (summed over workers)
};

// Where executed asynchronous requests are handed back.  A Coalescer has one of its own; a host with several event loops (the
// reference runs a monoio runtime per core, src/query_disk_index.rs:716-732) gives each loop its own queue, so that a request comes
// back to the loop that submitted it.  push() is lock-free and is the pusher's LAST access to the record; ring() moves the futex
// word (and the eventfd, if one was asked for) once per pass; take() hands records out in completion order, each exactly once.
class CompletionQueue {
  public:
    ~CompletionQueue();
    void push(DispatchReq* r);
    void ring();
    size_t take(DispatchReq** out, size_t max, int64_t timeout_us, const std::atomic<bool>* stop);
    int fd();          // an eventfd bumped by ring(); made on first call, closed with the queue
    void close_fd();
    void wake();       // wakes sleepers in take() without a completion (shutdown)
    // A worker's complete() pushes a pass's records and rings afterwards: between the two a poller may take the LAST outstanding
    // ticket and free the queue (mse.h allows that once none of its tickets is out).  The worker therefore pins the queue before its
    // first push and unpins after ring(); the destructor waits for the pins to drain.
    void pin() { pins_.fetch_add(1, std::memory_order_acq_rel); }
    void unpin() { pins_.fetch_sub(1, std::memory_order_release); }

  private:
    std::atomic<uint32_t> pins_{0};
    alignas(64) std::atomic<DispatchReq*> head_{nullptr};
    std::atomic<uint32_t> bell_{0};
    std::atomic<int> fd_{-1};
    std::mutex mu_;
    std::deque<DispatchReq*> ready_;
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
    // Asynchronous form (round 5): ONE host thread keeps thousands of requests in flight -- the device wants thousands of queries per
    // pass, and a sleeping OS thread per request is the wrong vehicle for that (4096 request threads on a 16-core CPU allowance spend
    // 25 us of CPU per request on being woken).  submit_async queues the record (owned by the caller, alive until it comes back) and
    // returns; completions() hands back up to `max` executed records, each exactly once, sleeping up to timeout_us for the first
    // (< 0: no limit); a shut-down handle answers what it still held with an error.  Any number of threads may call either.
    int submit_async(DispatchReq& r);
    size_t completions(DispatchReq** out, size_t max, int64_t timeout_us);
    // an eventfd (made on first call, owned by the handle) whose counter is bumped once per pass that completed asynchronous requests:
    // an event loop (epoll / io_uring) watches it, reads the 8-byte counter when it fires, then polls completions() until it returns 0
    int completion_fd();
    // (requests whose record names a CompletionQueue of its own -- DispatchReq::cq -- come back there instead)
    DispatchStats stats();
    size_t max_queries() const { return max_queries_; }
    uint32_t max_wait_us() const { return max_wait_us_.load(); }
    void set_max_wait_us(uint32_t us) { max_wait_us_.store(us); }   // takes effect from the next gather

  private:
    void loop(int index);
    void wake_loop(int index);         // a worker's companion: finishes large passes (the wake-ups) while the worker gathers the next
    void complete(std::vector<DispatchReq*>& batch);
    void enqueue(DispatchReq& r);      // the callers' side of submit / submit_async
    void drain();                      // inbox -> queue_ (the gatherer only)
    size_t target() const;             // queries the next pass waits for
    static constexpr uint32_t WAKE_SLOTS = 64, WAKE_RUN = 256;   // 256 consecutive arrivals share a slot
    struct alignas(64) WakeSlot { std::atomic<uint32_t> gen{0}; };
    const size_t max_queries_;
    std::atomic<uint32_t> max_wait_us_;
    RunFn run_;
    std::function<void()> on_start_;
    // ---- the callers' side: no lock.  A request is pushed onto a lock-free stack (callers only push, the gatherer takes the whole
    // stack at once and restores arrival order), the count of queries ever pushed is bumped, and the gatherer's bell is rung only by
    // the push that takes that count across `wake_at_` -- the first arrival after an idle spell, or the one that completes the
    // gatherer's target.  (With a mutex here, a thousand callers released by one pass formed a convoy on it: every contended
    // unlock a futex call and a context switch.)
    alignas(64) std::atomic<DispatchReq*> inbox_{nullptr};
    alignas(64) std::atomic<uint64_t> pushed_{0};     // queries ever pushed
    alignas(64) std::atomic<uint64_t> wake_at_{1};    // ring the bell when pushed_ reaches this
    std::atomic<uint32_t> bell_{0};                   // futex word the gatherer sleeps on
    std::atomic<uint64_t> arrivals_{0};
    std::atomic<uint64_t> taken_{0};                  // queries ever taken into a pass (pushed_ - taken_ = waiting now)
    std::atomic<bool> stop_{false};
    // Completion: a wake-up per request would be hundreds of futex calls per pass; ONE variable for all callers wakes every
    // sleeper of every other pass too.  A ring of futex words, 256 consecutive arrivals to a word (the queue is first in, first out,
    // so a pass touches few words): the worker sets each request's `done`, bumps the words it touched and wakes their sleepers; a
    // sleeper that was woken for somebody else's pass finds its own flag still clear and sleeps on the new value.
    WakeSlot wake_[WAKE_SLOTS];
    CompletionQueue own_cq_;          // executed asynchronous requests that name no queue of their own
    // ---- the workers' side.  mu_ is taken by workers (and stats()) only: it hands the gatherer's token around and guards the
    // statistics.  queue_ / queued_queries_ / drained_ belong to whoever holds the token.
    std::mutex mu_;
    std::condition_variable cv_worker_;
    bool gathering_ = false;          // one worker gathers at a time
    std::deque<DispatchReq*> queue_;
    size_t queued_queries_ = 0;
    uint64_t drained_ = 0;            // queries ever moved from the inbox to queue_
    std::atomic<size_t> expect_{1};
    std::atomic<bool> expected_returners_{false};
    std::atomic<int64_t> grace_until_ns_{0};   // steady_clock time since epoch, ns (0 = none)
    std::atomic<int> running_{0};              // passes executing now (several workers)
    bool hold_while_busy_ = true;              // a gatherer does not start a second pass by the wait budget while one is executing
    DispatchStats st_;
    std::vector<std::thread> workers_;
    // Waking several hundred sleepers is a futex call that costs about a microsecond per sleeper -- on the worker's critical path that
    // was as long as the pass itself.  A pass of more than WAKE_INLINE requests is handed to the worker's companion thread, which sets
    // the flags and rings the slots while the worker is already gathering the next pass.
    static constexpr size_t WAKE_INLINE = 96;
    struct Waker {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<DispatchReq*> todo;     // one pass at a time; the worker waits for the previous hand-over to be taken
        bool has = false, stop = false;
        std::thread th;
    };
    std::vector<std::unique_ptr<Waker>> wakers_;
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
