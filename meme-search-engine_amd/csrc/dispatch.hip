// Cross-thread query coalescer (dispatch.h) and its brute-force front: mse_dispatcher_* of include/mse.h.
#include "../../include/mse.h"
#include "dispatch.h"
#include "runtime.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <future>
#include <memory>
#include <new>
#include <climits>
#include <cerrno>
#include <linux/futex.h>
#include <sys/eventfd.h>
#include <sys/syscall.h>
#include <sched.h>
#include <unistd.h>
#include <time.h>

namespace mse {

// futex on a 32-bit atomic (std::atomic<uint32_t> is layout-compatible with uint32_t on this target)
static inline void futex_wait(std::atomic<uint32_t>* w, uint32_t seen) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
static inline void futex_wait_for(std::atomic<uint32_t>* w, uint32_t seen, int64_t ns) {
    struct timespec ts;
    ts.tv_sec = (time_t)(ns / 1000000000ll);
    ts.tv_nsec = (long)(ns % 1000000000ll);
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, &ts, nullptr, 0);   // relative timeout
}
static inline void futex_wake_all(std::atomic<uint32_t>* w) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

uint32_t default_wait_us(size_t n_rows, size_t row_bytes) {
    const double pass_us = (double)n_rows * (double)row_bytes / 4.0e6;   // bytes / (4 TB/s) in microseconds
    double w = pass_us / 10.0;
    if (w < 200.0) w = 200.0;
    if (w > 5000.0) w = 5000.0;
    return (uint32_t)w;
}

static thread_local int tl_worker_index = 0;
int Coalescer::worker_index() { return tl_worker_index; }

Coalescer::Coalescer(size_t max_queries, uint32_t max_wait_us, RunFn run, std::function<void()> on_thread_start, int n_workers)
    : max_queries_(max_queries ? max_queries : 1), max_wait_us_(max_wait_us), run_(std::move(run)),
      on_start_(std::move(on_thread_start)) {
    if (n_workers < 1) n_workers = 1;
    {
        const char* e = getenv("MSE_COALESCE_NO_HOLD");
        hold_while_busy_ = !(e && atoi(e));
    }
    for (int w = 0; w < n_workers; w++) {
        wakers_.emplace_back(new Waker());
        wakers_[w]->th = std::thread([this, w] { wake_loop(w); });
    }
    for (int w = 0; w < n_workers; w++) workers_.emplace_back([this, w] { loop(w); });
}

// flags first (a request's slot read before its flag is set: the record is gone once the flag is seen), then one wake-up per slot touched
void Coalescer::complete(std::vector<DispatchReq*>& batch) {
    uint64_t touched = 0;
    CompletionQueue* rung[8];
    int n_rung = 0;
    for (DispatchReq* r : batch) {
        if (r->async) {   // onto its completion queue: from the push on the record belongs to whoever takes it
            CompletionQueue* q = r->cq ? r->cq : &own_cq_;
            bool seen = false;
            for (int i = 0; i < n_rung; i++) seen |= rung[i] == q;
            const bool solo = !seen && n_rung >= 8;   // (more than eight queues in one pass: pinned, pushed and rung per request)
            if (!seen && !solo) {
                rung[n_rung++] = q;
                q->pin();          // before the first push: from that push on a poller may hold every ticket of the queue and free it
            }
            if (solo) q->pin();
            q->push(r);
            if (solo) {
                q->ring();
                q->unpin();
            }
            continue;
        }
        touched |= 1ull << r->slot;
        r->done.store(1, std::memory_order_release);
    }
    for (uint32_t sl = 0; sl < WAKE_SLOTS; sl++)
        if (touched >> sl & 1ull) {
            wake_[sl].gen.fetch_add(1, std::memory_order_release);
            futex_wake_all(&wake_[sl].gen);
        }
    for (int i = 0; i < n_rung; i++) {
        rung[i]->ring();
        rung[i]->unpin();
    }
}

size_t Coalescer::completions(DispatchReq** out, size_t max, int64_t timeout_us) { return own_cq_.take(out, max, timeout_us, &stop_); }
int Coalescer::completion_fd() { return own_cq_.fd(); }

CompletionQueue::~CompletionQueue() {
    while (pins_.load(std::memory_order_acquire)) sched_yield();   // a worker is between its push and its ring(): a few instructions
    close_fd();
}

void CompletionQueue::push(DispatchReq* r) {
    DispatchReq* head = head_.load(std::memory_order_relaxed);
    do {
        r->cnext = head;
    } while (!head_.compare_exchange_weak(head, r, std::memory_order_release, std::memory_order_relaxed));
}

void CompletionQueue::ring() {
    bell_.fetch_add(1, std::memory_order_release);
    futex_wake_all(&bell_);
    const int fd = fd_.load(std::memory_order_acquire);
    if (fd >= 0) {
        const uint64_t one = 1;
        (void)!write(fd, &one, sizeof one);   // (EAGAIN only at a counter of 2^64 - 2: the loop is awake anyway)
    }
}

void CompletionQueue::wake() {
    bell_.fetch_add(1, std::memory_order_release);
    futex_wake_all(&bell_);
}

void CompletionQueue::close_fd() {
    const int fd = fd_.exchange(-1);
    if (fd >= 0) (void)close(fd);
}

size_t CompletionQueue::take(DispatchReq** out, size_t max, int64_t timeout_us, const std::atomic<bool>* stop) {
    if (!out || max == 0) return 0;
    const bool timed = timeout_us >= 0;
    const int64_t deadline = timed ? std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() +
                                         timeout_us * 1000
                                   : 0;
    for (;;) {
        const uint32_t bell = bell_.load(std::memory_order_acquire);
        {
            std::lock_guard<std::mutex> lk(mu_);
            DispatchReq* h = head_.exchange(nullptr, std::memory_order_acquire);
            DispatchReq* rev = nullptr;   // the stack holds the newest first: back into completion order
            while (h) {
                DispatchReq* n = h->cnext;
                h->cnext = rev;
                rev = h;
                h = n;
            }
            for (DispatchReq* r = rev; r;) {
                DispatchReq* n = r->cnext;
                ready_.push_back(r);
                r = n;
            }
            size_t got = 0;
            while (got < max && !ready_.empty()) {
                out[got++] = ready_.front();
                ready_.pop_front();
            }
            if (got) return got;
        }
        if (stop && stop->load(std::memory_order_acquire)) return 0;
        if (!timed) {
            futex_wait(&bell_, bell);
            continue;
        }
        const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (now >= deadline) return 0;
        futex_wait_for(&bell_, bell, deadline - now);
    }
}

int CompletionQueue::fd() {
    int fd = fd_.load(std::memory_order_acquire);
    if (fd >= 0) return fd;
    const int made = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    if (made < 0) return fail(std::string("eventfd: ") + strerror(errno));
    int expected = -1;
    if (!fd_.compare_exchange_strong(expected, made, std::memory_order_acq_rel)) {   // another thread was first
        (void)close(made);
        return expected;
    }
    // requests that completed before the descriptor existed are announced now
    bool pending = head_.load(std::memory_order_acquire) != nullptr;
    if (!pending) {
        std::lock_guard<std::mutex> lk(mu_);
        pending = !ready_.empty();
    }
    if (pending) {
        const uint64_t one = 1;
        (void)!write(made, &one, sizeof one);
    }
    return made;
}

void Coalescer::wake_loop(int index) {
    Waker& w = *wakers_[index];
    std::vector<DispatchReq*> mine;
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.cv.wait(lk, [&] { return w.has || w.stop; });
        if (!w.has && w.stop) break;
        mine.swap(w.todo);
        w.has = false;
        lk.unlock();
        w.cv.notify_all();      // the worker may hand over its next pass
        complete(mine);
        mine.clear();
        lk.lock();
    }
}

Coalescer::~Coalescer() {
    stop_.store(true);
    bell_.fetch_add(1);
    futex_wake_all(&bell_);
    {
        std::lock_guard<std::mutex> lk(mu_);   // (a worker between its predicate and its wait holds mu_: it sees the flag afterwards)
    }
    cv_worker_.notify_all();
    for (std::thread& t : workers_)
        if (t.joinable()) t.join();
    for (auto& w : wakers_) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->stop = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();   // (a pass handed over before the stop is still completed: has is checked first)
    }
    // shutting down: nobody may stay blocked (submit refuses new requests once stop_ is set; what slipped in is answered here)
    drain();
    uint64_t touched = 0;
    for (DispatchReq* r : queue_) {
        r->rc = -1;
        r->err = "dispatcher closed while the request was queued";
        if (r->async) {   // handed back with the error through the queue it named; one that relied on this handle's own queue stays its
            // owner's, marked executed (that queue goes away with the handle)
            if (r->cq) {
                CompletionQueue* q = r->cq;
                q->push(r);
                q->ring();
            } else {
                r->done.store(1, std::memory_order_release);
            }
            continue;
        }
        touched |= 1ull << r->slot;
        r->done.store(1, std::memory_order_release);   // (r may be gone from here on)
    }
    queue_.clear();
    own_cq_.wake();   // anyone asleep in completions() sees stop_ and leaves
    own_cq_.close_fd();
    for (uint32_t sl = 0; sl < WAKE_SLOTS; sl++)
        if (touched >> sl & 1ull) {
            wake_[sl].gen.fetch_add(1, std::memory_order_release);
            futex_wake_all(&wake_[sl].gen);
        }
}

int Coalescer::submit_async(DispatchReq& r) {
    if (stop_.load(std::memory_order_acquire)) return fail("dispatcher is shutting down");
    r.async = true;
    r.done.store(0, std::memory_order_relaxed);
    r.slot = 0;
    enqueue(r);
    return 0;
}

int Coalescer::submit(DispatchReq& r) {
    if (stop_.load(std::memory_order_acquire)) return fail("dispatcher is shutting down");
    r.async = false;
    r.done.store(0, std::memory_order_relaxed);
    r.slot = (uint32_t)((arrivals_.fetch_add(1, std::memory_order_relaxed) / WAKE_RUN) % WAKE_SLOTS);
    enqueue(r);
    {
        std::atomic<uint32_t>& gen = wake_[r.slot].gen;
        for (;;) {
            const uint32_t seen = gen.load(std::memory_order_acquire);
            if (r.done.load(std::memory_order_acquire)) break;
            futex_wait(&gen, seen);   // returns at once if the word has moved on since it was read
        }
    }
    if (r.rc) set_error(r.err.empty() ? std::string("search failed") : r.err);
    return r.rc;
}

void Coalescer::enqueue(DispatchReq& r) {
    r.t_arrive = std::chrono::steady_clock::now();
    const size_t r_nq = r.nq;   // (an asynchronous record may be executed, handed back and freed as soon as it is in the inbox)
    DispatchReq* head = inbox_.load(std::memory_order_relaxed);
    do {
        r.next = head;
    } while (!inbox_.compare_exchange_weak(head, &r, std::memory_order_release, std::memory_order_relaxed));
    // the bell: only the push that carries the count across the gatherer's mark (both sides sequentially consistent: either this
    // thread sees the mark the gatherer published, or the gatherer sees this push when it re-reads the count before sleeping)
    const uint64_t before = pushed_.fetch_add(r_nq, std::memory_order_seq_cst), after = before + r_nq;
    const uint64_t mark = wake_at_.load(std::memory_order_seq_cst);
    if (before < mark && after >= mark) {
        bell_.fetch_add(1, std::memory_order_release);
        futex_wake_all(&bell_);
    }
}

// How many queries the next pass waits for: everything expected if one pass holds it; otherwise the expected population in EQUAL
// passes -- 512 closed-loop callers against a 320-query pass are served as 256 + 256 (2 x 58 ms), not 320 + 192 (70 + 48 ms and a
// gather that keeps waiting for the stragglers of the large pass): 3.8-3.9 k -> queries/s of profiles/r04_bench_default.json.
size_t Coalescer::target() const {
    const size_t expect = expect_.load(std::memory_order_acquire);
    if (expect <= max_queries_) return expect;
    const size_t passes = (expect + max_queries_ - 1) / max_queries_;
    return (expect + passes - 1) / passes;
}

DispatchStats Coalescer::stats() {
    std::lock_guard<std::mutex> lk(mu_);
    return st_;
}

// the whole inbox at once, back into arrival order, behind what is already queued (the gatherer only)
void Coalescer::drain() {
    DispatchReq* h = inbox_.exchange(nullptr, std::memory_order_acquire);
    DispatchReq* rev = nullptr;
    size_t nq = 0;
    while (h) {
        DispatchReq* n = h->next;
        h->next = rev;
        rev = h;
        h = n;
    }
    for (DispatchReq* r = rev; r; r = r->next) {
        queue_.push_back(r);
        nq += r->nq;
    }
    queued_queries_ += nq;
    drained_ += nq;
}

static inline int64_t steady_ns(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(t.time_since_epoch()).count();
}

void Coalescer::loop(int index) {
    tl_worker_index = index;
    if (on_start_) on_start_();
    std::vector<DispatchReq*> batch;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
        cv_worker_.wait(lk, [&] { return stop_.load() || !gathering_; });
        if (stop_.load()) break;
        gathering_ = true;
        lk.unlock();
        // ---- the gatherer: sleeps on the bell, never on a lock the callers touch ----
        // until the first request is there ...
        bool by_deadline = false, held = false;
        auto sleep_until_mark = [&](uint64_t mark, bool timed, int64_t deadline_ns) -> bool {   // false: the deadline passed
            const uint32_t b = bell_.load(std::memory_order_acquire);
            wake_at_.store(mark, std::memory_order_seq_cst);
            if (pushed_.load(std::memory_order_seq_cst) >= mark || stop_.load()) return true;   // it came while the mark was being set
            if (!timed) { futex_wait(&bell_, b); return true; }
            const int64_t now = steady_ns(std::chrono::steady_clock::now());
            if (now >= deadline_ns) return false;
            futex_wait_for(&bell_, b, deadline_ns - now);
            return true;
        };
        for (;;) {
            drain();
            if (!queue_.empty() || stop_.load()) break;
            (void)sleep_until_mark(drained_ + 1, false, 0);
        }
        if (stop_.load()) { lk.lock(); gathering_ = false; break; }
        // ... then until the target is waiting (dispatch.h), or the oldest request is max_wait old -- but never before the callers
        // just answered had their grace period to come back.  expect_ / the grace period may move while this waits (another worker's
        // pass ending rings the bell): both are re-read every round.
        const bool had_expected = expected_returners_.load();
        for (;;) {
            drain();
            const size_t tgt = target();
            if (queued_queries_ >= tgt || stop_.load()) break;
            int64_t deadline = steady_ns(queue_.front()->t_arrive) + (int64_t)max_wait_us_.load() * 1000;
            deadline = std::max(deadline, grace_until_ns_.load(std::memory_order_acquire));
            // While another worker's pass is executing the device is busy anyway: a second pass started by the wait budget would only
            // be a smaller one (4096 requests in flight ran as passes of 400-700 queries at 1.6 us per query; a pass of 2048 costs 0.8).
            // The gatherer then waits for its target or for that pass to end (its end rings the bell), whichever comes first.
            if (hold_while_busy_) {
                // (the bell word is read BEFORE the count of executing passes: a pass that ends after that read rings a bell this wait
                // still sees -- its decrement comes before its ring)
                // The hold is bounded: a slow pass (the grow-and-repeat path, a failed shared submission retried request by request)
                // must not stall every queued request while the other workers idle -- after 8 wait budgets past the deadline the
                // gatherer goes on by the ordinary rule.
                const uint32_t b = bell_.load(std::memory_order_acquire);
                const int64_t now = steady_ns(std::chrono::steady_clock::now());
                const int64_t hold_end = deadline + 8 * (int64_t)max_wait_us_.load() * 1000;
                if (running_.load(std::memory_order_acquire) > 0 && now < hold_end) {
                    const uint64_t mark = drained_ + (tgt - queued_queries_);
                    wake_at_.store(mark, std::memory_order_seq_cst);
                    if (pushed_.load(std::memory_order_seq_cst) < mark && !stop_.load()) futex_wait_for(&bell_, b, hold_end - now);
                    held = true;
                    continue;
                }
            }
            // (a deadline that passed while the gatherer was HOLDING says nothing about returners that did not come back)
            if (!sleep_until_mark(drained_ + (tgt - queued_queries_), true, deadline)) { by_deadline = !held; drain(); break; }
        }
        if (stop_.load()) { lk.lock(); gathering_ = false; break; }
        batch.clear();
        size_t nq = 0;
        // more callers than one pass holds: equal shares (target()), not a full pass and a remainder
        const size_t cap = expect_.load() > max_queries_ ? target() : max_queries_;
        while (!queue_.empty()) {
            DispatchReq* r = queue_.front();
            if (!batch.empty() && nq + r->nq > cap) break;   // a request larger than a pass goes alone
            batch.push_back(r);
            nq += r->nq;
            queue_.pop_front();
        }
        queued_queries_ -= nq;
        taken_.fetch_add(nq, std::memory_order_release);
        lk.lock();
        gathering_ = false;                       // what is left (and what arrives) is the next gatherer's
        lk.unlock();
        if (workers_.size() > 1) cv_worker_.notify_one();
        const auto t_run = std::chrono::steady_clock::now();
        running_.fetch_add(1, std::memory_order_acq_rel);
        run_(batch);
        running_.fetch_sub(1, std::memory_order_acq_rel);
        const auto t_end = std::chrono::steady_clock::now();
        const uint64_t run_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_end - t_run).count();
        // Next target: what queued up during this pass PLUS the callers answered now -- closed-loop callers (a thread per core,
        // one request at a time) are back within microseconds, and without counting them T callers settle into two groups of T/2
        // taking turns.  A CALLER comes back, whatever it asked for: a request of 256 queries counts as one returner, not as 256
        // (a lone one-query caller arriving after a batch call would otherwise wait for 255 more that never come).  If the gather
        // that just ended had counted on returners and ran into its deadline instead (they did not come back: open-loop arrivals,
        // or callers that left), the next one does not count on them; the one after tries again.
        const bool expect_returners = !(by_deadline && had_expected);
        expected_returners_.store(expect_returners);
        // waiting now: pushed and not yet part of any pass
        const uint64_t taken_now = taken_.load(std::memory_order_acquire), pushed_now = pushed_.load(std::memory_order_acquire);
        const size_t waiting = (size_t)(pushed_now > taken_now ? pushed_now - taken_now : 0);
        {
            std::lock_guard<std::mutex> g(mu_);
            st_.run_us += run_us;
            st_.passes++;
            st_.requests += batch.size();
            st_.queries += nq;
            st_.max_pass_queries = std::max<uint64_t>(st_.max_pass_queries, nq);
            if (by_deadline) st_.deadline_fires++;
        }
        expect_.store(std::max<size_t>(1, waiting + (expect_returners ? batch.size() : 0)), std::memory_order_release);
        const uint32_t grace_us = std::min<uint32_t>(max_wait_us_.load(), 1000);
        grace_until_ns_.store(expect_returners ? steady_ns(t_end) + (int64_t)grace_us * 1000 : 0, std::memory_order_release);
        // completion: inline for a small pass; a large one goes to the companion thread (its wake-ups take as long as a pass)
        if (batch.size() <= WAKE_INLINE) {
            complete(batch);
        } else {
            Waker& wk = *wakers_[index];
            std::unique_lock<std::mutex> wl(wk.mu);
            wk.cv.wait(wl, [&] { return !wk.has; });
            wk.todo.swap(batch);
            wk.has = true;
            wl.unlock();
            wk.cv.notify_all();
        }
        if (workers_.size() > 1) {   // a gatherer waiting for its target re-reads expect_ / the grace period
            bell_.fetch_add(1, std::memory_order_release);
            futex_wake_all(&bell_);
        }
        lk.lock();
    }
    lk.unlock();
    cv_worker_.notify_all();
}

}  // namespace mse

using namespace mse;

// Brute-force front: T threads with one f16 query each (or a few) share one pass over the base rows.
struct mse_dispatcher {
    const mse_base* base = nullptr;
    mse_searcher* s = nullptr;          // made on the worker thread (its stream lives on the base's device)
    std::string start_error;
    void* pin = nullptr;                // pinned staging: queries up, [scores | ids] down
    size_t pin_cap = 0;
    DevBuf q_dev, out_dev;
    std::unique_ptr<Coalescer> co;
    std::atomic<uint64_t> retried_alone{0};
    std::atomic<uint32_t> fail_shared{0};   // test hook: this many shared passes fail before they start
};

namespace {

// one engine call for a contiguous group of requests; on success every request has its rows
int run_group(mse_dispatcher* D, DispatchReq* const* reqs, size_t n_req) {
    const mse_base* b = D->base;
    const size_t d = b->d;
    size_t total = 0, kmax = 0;
    for (size_t i = 0; i < n_req; i++) {
        total += reqs[i]->nq;
        kmax = std::max(kmax, reqs[i]->k);
    }
    if (total == 0 || kmax == 0) return 0;
    if (!D->s) return fail(D->start_error.empty() ? std::string("dispatcher has no searcher") : D->start_error);
    mse_searcher* s = D->s;
    hipStream_t st = s->stream;
    const size_t in_bytes = total * d * 2, sc_bytes = total * kmax * 8, id_bytes = total * kmax * 4;
    const size_t out_bytes = sc_bytes + id_bytes;
    if (D->pin_cap < std::max(in_bytes, out_bytes)) {
        if (D->pin) (void)hipHostFree(D->pin);
        D->pin = nullptr;
        D->pin_cap = 0;
        const size_t want = std::max<size_t>(2 * std::max(in_bytes, out_bytes), (size_t)1 << 20);
        MSE_HIP_TRY(hipHostMalloc(&D->pin, want, hipHostMallocDefault));
        D->pin_cap = want;
    }
    if (D->q_dev.ensure(in_bytes) || D->out_dev.ensure(out_bytes)) return -1;
    char* p = static_cast<char*>(D->pin);
    for (size_t i = 0, o = 0; i < n_req; i++) {
        memcpy(p + o, reqs[i]->queries, reqs[i]->nq * d * 2);
        o += reqs[i]->nq * d * 2;
    }
    MSE_HIP_TRY(hipMemcpyAsync(D->q_dev.p, D->pin, in_bytes, hipMemcpyHostToDevice, st));
    // a pass of the matrix-core scan costs less than the exact-order pass once the rows no longer fit the caches, whatever the
    // query count (40 ms against 54 ms at 1e8 rows); below that the exact pass has the shorter tail.  Same answers either way.
    const int mode = (total > 8 || b->n >= ((size_t)1 << 22)) ? MSE_MODE_MFMA : MSE_MODE_EXACT;
    int64_t* sc_dev = D->out_dev.as<int64_t>();
    uint32_t* id_dev = reinterpret_cast<uint32_t*>(D->out_dev.as<char>() + sc_bytes);
    if (mse_bruteforce_topk_f16_dev(s, D->q_dev.p, total, kmax, mode, 0, sc_dev, id_dev)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(D->pin, D->out_dev.p, out_bytes, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    const int64_t* sc = reinterpret_cast<const int64_t*>(p);
    const uint32_t* id = reinterpret_cast<const uint32_t*>(p + sc_bytes);
    // the k best of a caller are the first k of the kmax best: the order (score descending, id ascending) is total
    for (size_t i = 0, row = 0; i < n_req; i++) {
        DispatchReq* r = reqs[i];
        for (size_t q = 0; q < r->nq; q++, row++) {
            memcpy(static_cast<int64_t*>(r->out_a) + q * r->k, sc + row * kmax, r->k * 8);
            memcpy(static_cast<uint32_t*>(r->out_b) + q * r->k, id + row * kmax, r->k * 4);
        }
    }
    return 0;
}

void run_batch(mse_dispatcher* D, std::vector<DispatchReq*>& batch) {
    bool injected = false;
    if (batch.size() > 1 && D->fail_shared.load() > 0) { D->fail_shared--; injected = true; fail("injected failure of a shared pass (test hook)"); }
    if (!injected && run_group(D, batch.data(), batch.size()) == 0) {
        for (DispatchReq* r : batch) r->rc = 0;
        return;
    }
    if (batch.size() == 1) {
        batch[0]->rc = -1;
        batch[0]->err = mse_last_error();
        return;
    }
    // the shared pass failed: every request is repeated on its own, so that a caller only ever sees its own failure
    for (DispatchReq* r : batch) {
        D->retried_alone++;
        r->rc = run_group(D, &r, 1);
        if (r->rc) r->err = mse_last_error();
    }
}

}  // namespace

extern "C" {

mse_dispatcher* mse_dispatcher_new(const mse_base* b, size_t max_queries_per_pass, uint32_t max_wait_us) {
    if (!b) { fail("null base"); return nullptr; }
    mse_dispatcher* D = new (std::nothrow) mse_dispatcher();
    if (!D) { fail("out of host memory"); return nullptr; }
    D->base = b;
    const size_t mq = max_queries_per_pass ? max_queries_per_pass : (size_t)mfma_query_tile((int)b->d);
    const uint32_t wait = max_wait_us ? max_wait_us : default_wait_us(b->n, b->d * 2);
    const int device = b->device;
    std::promise<void> started;
    D->co.reset(new Coalescer(
        mq, wait, [D](std::vector<DispatchReq*>& batch) { run_batch(D, batch); },
        [D, device, &started] {
            // HIP's current device is per thread: the worker lives on the device that holds the rows
            if (hipSetDevice(device) != hipSuccess) D->start_error = "dispatcher: hipSetDevice failed";
            else if (!(D->s = mse_searcher_new(D->base))) D->start_error = mse_last_error();
            started.set_value();
        }));
    started.get_future().wait();
    if (!D->s) {
        const std::string why = D->start_error;
        mse_dispatcher_free(D);
        fail(why.empty() ? std::string("dispatcher: worker failed to start") : why);
        return nullptr;
    }
    return D;
}

void mse_dispatcher_free(mse_dispatcher* D) {
    if (!D) return;
    D->co.reset();   // joins the worker
    if (D->s) mse_searcher_free(D->s);
    if (D->pin) (void)hipHostFree(D->pin);
    delete D;
}

int mse_dispatcher_topk_f16(mse_dispatcher* D, const uint16_t* queries, size_t nq, size_t k, int64_t* scores, uint32_t* ids) {
    if (!D) return fail("null dispatcher");
    if (nq == 0 || k == 0) return 0;
    // argument errors never enter the queue: they belong to this caller alone
    if (!queries || !scores || !ids) return fail("null argument");
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    DispatchReq r;
    r.queries = queries;
    r.nq = nq;
    r.k = k;
    r.out_a = scores;
    r.out_b = ids;
    return D->co->submit(r);
}

int mse_dispatcher_stats(mse_dispatcher* D, uint64_t out[6]) {
    if (!D || !out) return fail("null argument");
    const DispatchStats st = D->co->stats();
    out[0] = st.queries;
    out[1] = st.requests;
    out[2] = st.passes;
    out[3] = st.max_pass_queries;
    out[4] = st.deadline_fires;
    out[5] = D->retried_alone.load();
    return 0;
}

mse_searcher* mse_dispatcher_searcher(mse_dispatcher* D) { return D ? D->s : nullptr; }

// test hook, no device needed: `threads` host threads x `rounds` one-query requests through a Coalescer whose "pass" answers
// request payload p with 2 p + 1 and fails (for that request alone) every payload divisible by 97.  *mismatches = requests that
// got somebody else's answer, a wrong status, or a missing error text.
int mse_debug_coalescer_selftest(int threads, int rounds, uint32_t max_queries, uint32_t max_wait_us, uint64_t stats_out[6],
                                 uint64_t* mismatches) {
    return mse_debug_coalescer_selftest_workers(threads, rounds, max_queries, max_wait_us, 1, stats_out, mismatches);
}

int mse_debug_coalescer_selftest_workers(int threads, int rounds, uint32_t max_queries, uint32_t max_wait_us, int workers, uint64_t stats_out[6],
                                         uint64_t* mismatches) {
    if (threads <= 0 || rounds <= 0 || workers <= 0 || !stats_out || !mismatches) return fail("bad argument");
    std::atomic<uint64_t> wrong_worker{0};
    Coalescer co(max_queries ? max_queries : 256, max_wait_us ? max_wait_us : 200,
                 [&wrong_worker, workers](std::vector<DispatchReq*>& batch) {
                     if (Coalescer::worker_index() < 0 || Coalescer::worker_index() >= workers) wrong_worker++;
                     for (DispatchReq* r : batch) {
                         const uint64_t p = *static_cast<const uint64_t*>(r->queries);
                         if (p % 97 == 0) { r->rc = -1; r->err = "payload " + std::to_string(p) + " refused"; }
                         else { *static_cast<uint64_t*>(r->out_a) = 2 * p + 1; r->rc = 0; }
                     }
                 },
                 nullptr, workers);
    std::atomic<uint64_t> bad{0};
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; t++)
        ts.emplace_back([&, t] {
            for (int r = 0; r < rounds; r++) {
                uint64_t p = (uint64_t)t * 1000003ull + (uint64_t)r, out = 0;
                DispatchReq q;
                q.queries = &p; q.nq = 1; q.k = 1; q.out_a = &out;
                const int rc = co.submit(q);
                if (p % 97 == 0) { if (rc == 0 || std::string(mse_last_error()).find(std::to_string(p)) == std::string::npos) bad++; }
                else if (rc != 0 || out != 2 * p + 1) bad++;
            }
        });
    for (std::thread& th : ts) th.join();
    const DispatchStats st = co.stats();
    stats_out[0] = st.queries; stats_out[1] = st.requests; stats_out[2] = st.passes; stats_out[3] = st.max_pass_queries;
    stats_out[4] = st.deadline_fires; stats_out[5] = 0;
    *mismatches = bad.load() + wrong_worker.load();
    return 0;
}

// test hook, no device needed: the asynchronous side of the queue.  `async_threads` threads each keep `window` one-query records in
// flight (n_requests each, collecting whatever completes -- their own records or another thread's) while `sync_threads` blocking
// callers run beside them through the same Coalescer; the stand-in pass is the one above.  *mismatches = records handed back twice or
// never, wrong answers / statuses / error texts.
int mse_debug_coalescer_selftest_async(int async_threads, int window, int n_requests, int sync_threads, uint32_t max_queries, int workers,
                                       int own_queues, uint64_t stats_out[6], uint64_t* mismatches) {
    if (async_threads <= 0 || window <= 0 || n_requests <= 0 || sync_threads < 0 || workers <= 0 || !stats_out || !mismatches) return fail("bad argument");
    struct Rec {
        DispatchReq r;
        uint64_t p = 0, out = 0;
        std::atomic<int> handed{0};
        int thread = 0;
    };
    std::vector<std::unique_ptr<CompletionQueue>> queues;   // own_queues: one per asynchronous thread (declared first: outlives the handle)
    for (int t = 0; own_queues && t < async_threads; t++) queues.emplace_back(new CompletionQueue());
    const size_t total = (size_t)async_threads * (size_t)n_requests;
    std::vector<std::unique_ptr<Rec>> recs(total);
    for (size_t i = 0; i < total; i++) {
        recs[i].reset(new Rec());
        recs[i]->p = 5000000ull + i;
        recs[i]->r.queries = &recs[i]->p; recs[i]->r.nq = 1; recs[i]->r.k = 1; recs[i]->r.out_a = &recs[i]->out; recs[i]->r.owner = recs[i].get();
        recs[i]->thread = (int)(i / (size_t)n_requests);
        if (own_queues) recs[i]->r.cq = queues[recs[i]->thread].get();
    }
    std::atomic<uint64_t> bad{0}, collected{0};
    std::atomic<int64_t> in_flight{0};
    {
        Coalescer co(max_queries ? max_queries : 256, 200,
                     [](std::vector<DispatchReq*>& batch) {
                         for (DispatchReq* r : batch) {
                             const uint64_t p = *static_cast<const uint64_t*>(r->queries);
                             if (p % 97 == 0) { r->rc = -1; r->err = "payload " + std::to_string(p) + " refused"; }
                             else { *static_cast<uint64_t*>(r->out_a) = 2 * p + 1; r->rc = 0; }
                         }
                     },
                     nullptr, workers);
        std::vector<std::thread> ts;
        for (int t = 0; t < async_threads; t++)
            ts.emplace_back([&, t] {
                size_t next = 0, mine_back = 0;
                int64_t mine_out = 0;
                DispatchReq* got[64];
                while (own_queues ? mine_back < (size_t)n_requests : collected.load() < total) {
                    // shared queue: completions go to whichever thread asks, so the count in flight is kept for all of them;
                    // own queues: every thread keeps its own window and gets back exactly what it submitted
                    while (next < (size_t)n_requests && (own_queues ? mine_out < window : in_flight.load() < (int64_t)window * async_threads)) {
                        in_flight.fetch_add(1);
                        mine_out++;
                        if (co.submit_async(recs[(size_t)t * n_requests + next]->r)) bad++;
                        next++;
                    }
                    const size_t n = own_queues ? queues[t]->take(got, 64, 2000, nullptr) : co.completions(got, 64, 2000);
                    mine_out -= (int64_t)n;
                    mine_back += n;
                    for (size_t i = 0; i < n; i++) {
                        Rec* rc = static_cast<Rec*>(got[i]->owner);
                        if (own_queues && rc->thread != t) bad++;   // somebody else's record in this thread's queue
                        if (rc->handed.fetch_add(1) != 0) bad++;
                        if (rc->p % 97 == 0) { if (rc->r.rc == 0 || rc->r.err.find(std::to_string(rc->p)) == std::string::npos) bad++; }
                        else if (rc->r.rc != 0 || rc->out != 2 * rc->p + 1) bad++;
                    }
                    in_flight.fetch_sub((int64_t)n);
                    collected.fetch_add(n);
                }
            });
        for (int t = 0; t < sync_threads; t++)
            ts.emplace_back([&, t] {
                for (int r = 0; r < 200; r++) {
                    uint64_t p = 900000000ull + (uint64_t)t * 1000ull + (uint64_t)r, out = 0;
                    DispatchReq q;
                    q.queries = &p; q.nq = 1; q.k = 1; q.out_a = &out;
                    const int rc = co.submit(q);
                    if (p % 97 == 0) { if (rc == 0) bad++; }
                    else if (rc != 0 || out != 2 * p + 1) bad++;
                }
            });
        for (std::thread& th : ts) th.join();
        const DispatchStats st = co.stats();
        stats_out[0] = st.queries; stats_out[1] = st.requests; stats_out[2] = st.passes; stats_out[3] = st.max_pass_queries;
        stats_out[4] = st.deadline_fires; stats_out[5] = collected.load();
    }
    for (size_t i = 0; i < total; i++)
        if (recs[i]->handed.load() != 1) bad++;
    *mismatches = bad.load();
    return 0;
}

int mse_debug_dispatcher_fail_shared(mse_dispatcher* D, uint32_t n_passes) {
    if (!D) return fail("null dispatcher");
    D->fail_shared.store(n_passes);
    return 0;
}

}  // extern "C"
