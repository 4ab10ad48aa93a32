// host_stream_pool.h -- part of dpdf_model.hip (included there, in this order; one translation unit): C ABI: native coalescing of independent submitters (dpdf_streams_submit*): rounds, leader, windows.

// ------------------------------------------------------------------------------------------------
// Native coalescing of INDEPENDENT submitters (the reference's pattern: N StreamEnhancer objects, each fed by its own caller
// whenever it has a chunk, package/src/dpdfnet/stream.py:13-72, 74-165).  Host threads submit k hops for one slot each; the
// first submitter of a round leads it: it waits -- at most the window, and only until every slot in use has queued -- then
// issues ONE masked device call for everybody in the round.  Submitters write their samples straight into the round's pinned,
// GPU-visible input block and read their result straight out of its output block (both in parallel, outside the lock);
// two round buffers alternate, so the next round fills while this one is on the GPU.  A round is homogeneous in k: a request
// with another hop count waits for the open round to fire and opens / joins the next one.
// ------------------------------------------------------------------------------------------------
struct StreamPoolC {
    struct Round {
        enum { OPEN = 0, FIRING = 1, DONE = 2 };
        std::atomic<int> state{OPEN};                 // written under `mu`; pool_collect's bounded poll reads it without
        int k = 0, n_queued = 0, n_regular_queued = 0, copies_pending = 0, readers_left = 0, rc = 0;
        bool has_leader = false;
        std::atomic<long> id{0};
        std::vector<unsigned char> active;
        std::string err;
        float* pin_in = nullptr; float* pin_out = nullptr; size_t cap = 0;       // floats
    };
    std::mutex mu, exec_mu;
    std::condition_variable cv;                       // every state change (arrivals, copies finished, rounds done / recycled)
    Round rd[2];
    long open_id = 0;                                 // id of the round that takes submissions (buffer open_id & 1)
    std::vector<unsigned char> in_use; int n_in_use = 0;
    std::vector<long> slot_round;                     // round id of the slot's outstanding request (-1: none)
    double window_s = 2e-4;                           // wait for in-use slots that did NOT ride in the previous round
    double regular_window_s = 2e-3;                   // ... for the ones that did (they are feeding the pool hop after hop: worth waiting for)
    double spin_s = 0.0;                              // waiters poll (no futex sleep) for up to this long before they block on `cv`
    std::vector<unsigned char> prev_active; int n_prev_in_use = 0;     // who rode in the last round that fired (and is still in use)
    std::atomic<int> arrivals{0};                     // bumped by every join: what a polling leader watches
    int n_cv_waiters = 0;                             // threads asleep on `cv` (nobody: state changes skip the notify)
    int n_blocked = 0;                                // submitters waiting in pool_join (another hop count than the open round's, or its buffer not recycled yet): they cannot join the open round
    long device_calls = 0, rounds = 0;
    // where a round's time goes (dpdf_streams_pool_timing): leader waiting for the others | device call | end of a call -> next call
    double t_wait = 0, t_call = 0, t_gap = 0; std::chrono::steady_clock::time_point last_done{};
    // Is more than one host thread feeding the pool?  A leader that is the only recent submitter does not wait for others.
    std::thread::id last_tid{}; std::chrono::steady_clock::time_point other_seen{};
};
// The pool's critical sections are tens of instructions long; a std::mutex that is found locked puts the caller to sleep in the
// kernel (tens of microseconds, per feeder thread and round).  Try for a few microseconds first.
static inline void pool_lock(std::unique_lock<std::mutex>& lk) {
    for (int i = 0; i < 4000; ++i) {
        if (lk.try_lock()) return;
        __builtin_ia32_pause();
    }
    lk.lock();
}
static StreamPoolC* pool_of(dpdf_streams* s) {
    static std::mutex g_mu;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!s->pool) {
        s->pool = new StreamPoolC();
        s->pool->in_use.assign(s->S, 0);
        s->pool->slot_round.assign(s->S, -1);
        s->pool->prev_active.assign(s->S, 0);
        for (int b = 0; b < 2; ++b) { s->pool->rd[b].active.assign(s->S, 0); s->pool->rd[b].id = b; }
    }
    return s->pool;
}
static void pool_destroy(dpdf_streams* s) {
    if (!s->pool) return;
    for (int b = 0; b < 2; ++b) {
        if (s->pool->rd[b].pin_in) (void)hipHostFree(s->pool->rd[b].pin_in);
        if (s->pool->rd[b].pin_out) (void)hipHostFree(s->pool->rd[b].pin_out);
    }
    delete s->pool; s->pool = nullptr;
}
extern "C" int dpdf_streams_pool_config(dpdf_streams* s, double window_s) {
    if (!s || !(window_s >= 0.0)) return set_err(DPDF_E_INVALID, "bad argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    P->window_s = window_s;
    P->regular_window_s = window_s > 2e-3 ? window_s : 2e-3;          // (the default; dpdf_streams_pool_tune sets it on its own)
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_tune(dpdf_streams* s, double window_s, double regular_window_s, double spin_s) {
    if (!s || !(window_s >= 0.0) || !(regular_window_s >= 0.0) || !(spin_s >= 0.0)) return set_err(DPDF_E_INVALID, "bad argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    P->window_s = window_s; P->regular_window_s = regular_window_s > window_s ? regular_window_s : window_s; P->spin_s = spin_s;
    return DPDF_OK;
}
extern "C" int dpdf_streams_slot_use(dpdf_streams* s, int slot, int in_use) {
    if (!s || slot < 0 || slot >= s->S) return set_err(DPDF_E_INVALID, "bad slot");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    if (in_use && !P->in_use[slot]) { P->in_use[slot] = 1; ++P->n_in_use; }
    else if (!in_use && P->in_use[slot]) {
        P->in_use[slot] = 0; --P->n_in_use;
        if (P->prev_active[slot]) { P->prev_active[slot] = 0; --P->n_prev_in_use; }
        P->arrivals.fetch_add(1, std::memory_order_release); P->cv.notify_all();
    }
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_stats(dpdf_streams* s, long* device_calls, long* rounds) {
    if (!s) return set_err(DPDF_E_INVALID, "null streams");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    if (device_calls) *device_calls = P->device_calls;
    if (rounds) *rounds = P->rounds;
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_timing(dpdf_streams* s, double* out3) {
    if (!s || !out3) return set_err(DPDF_E_INVALID, "null argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    out3[0] = P->t_wait; out3[1] = P->t_call; out3[2] = P->t_gap;
    return DPDF_OK;
}
// join the open round with k hops for each of `n` slots of ONE caller (idx[] picks the requests out of the caller's arrays), copying
// the samples in; *lead_id >= 0: this caller has become that round's leader.  The whole group joins ONE round under one acquisition of
// the pool's lock (a contended std::mutex puts the loser to sleep in the kernel: per-slot locking cost 20-40 us per feeder thread and
// round with four feeders).
static int pool_join(dpdf_streams* s, StreamPoolC* P, int n, const int* idx, const int* slots, const float* const* in_rows, int k, long* lead_id) {
    const dpdf_dims& d = s->m->d;
    const size_t row = (size_t)k * d.hop;
    std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
    for (int j = 0; j < n; ++j)
        if (P->slot_round[slots[idx[j]]] >= 0) return set_err(DPDF_E_STATE, "slot %d already has a request in flight", slots[idx[j]]);
    StreamPoolC::Round* R;
    bool blocked = false;
    for (;;) {
        R = &P->rd[P->open_id & 1];
        // the buffer of the open round is free once the readers of the round that used it before are through; and a round takes
        // one hop count only
        if (R->state == StreamPoolC::Round::OPEN && R->id == P->open_id && (R->n_queued == 0 || R->k == k)) break;
        if (!blocked) {        // said ONCE: the open round's leader counts us as "cannot come" (several blocked submitters that re-announced themselves on every wake-up kept waking each other until the round fired)
            blocked = true;
            P->n_blocked += n; P->arrivals.fetch_add(1, std::memory_order_release); P->cv.notify_all();
        }
        ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters;
    }
    if (blocked) P->n_blocked -= n;
    if (R->n_queued == 0) {
        R->k = k;
        const size_t need = (size_t)s->S * row;
        if (need > R->cap) {                           // (empty round, buffer idle: nobody reads or writes it)
            if (hipSetDevice(s->m->device) != hipSuccess) return set_err(DPDF_E_RUNTIME, "hipSetDevice failed");
            if (R->pin_in) (void)hipHostFree(R->pin_in);
            if (R->pin_out) (void)hipHostFree(R->pin_out);
            R->pin_in = R->pin_out = nullptr; R->cap = 0;
            if (hipHostMalloc((void**)&R->pin_in, need * sizeof(float), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void**)&R->pin_out, need * sizeof(float), hipHostMallocDefault) != hipSuccess)
                return set_err(DPDF_E_RUNTIME, "hipHostMalloc of the pool's round buffers failed");
            R->cap = need;
        }
    }
    const auto tid = std::this_thread::get_id();
    if (P->last_tid != std::thread::id{} && P->last_tid != tid) P->other_seen = std::chrono::steady_clock::now();
    P->last_tid = tid;
    for (int j = 0; j < n; ++j) {
        const int slot = slots[idx[j]];
        R->active[slot] = 1;
        if (P->prev_active[slot]) ++R->n_regular_queued;
        P->slot_round[slot] = R->id;
    }
    R->n_queued += n; ++R->copies_pending;
    if (!R->has_leader) { R->has_leader = true; *lead_id = R->id; }
    float* base = R->pin_in;
    lk.unlock();
    for (int j = 0; j < n; ++j) memcpy(base + (size_t)slots[idx[j]] * row, in_rows[idx[j]], row * sizeof(float));
    pool_lock(lk);
    --R->copies_pending;
    P->arrivals.fetch_add(1, std::memory_order_release);
    if (P->n_cv_waiters) P->cv.notify_all();
    return DPDF_OK;
}
static void pool_lead(dpdf_streams* s, StreamPoolC* P, long id, bool no_window) {
    StreamPoolC::Round* R = &P->rd[id & 1];
    const auto t_lead0 = std::chrono::steady_clock::now();
    {
        using clk = std::chrono::steady_clock;
        std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
        const auto t0 = clk::now();
        const bool others = P->other_seen != clk::time_point{} && t0 - P->other_seen < std::chrono::seconds(1);
        if (!no_window && others && P->window_s > 0) {
            auto dur = [](double sec) { return std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(sec)); };
            const auto t_short = t0 + dur(P->window_s), t_long = t0 + dur(P->regular_window_s), t_spin = t0 + dur(P->spin_s);
            // (a slot has at most one request per round: once as many are queued as slots are in use, nobody else can come.)
            // Two windows: slots that rode in the previous round are being fed hop after hop -- their submitters are on their way
            // back (a wake-up, some host code), and a round fired without them costs everybody a second device call: the leader
            // waits `regular_window_s` for those; for in-use slots that sat the last round out only `window_s`.
            for (;;) {
                if (R->n_queued + P->n_blocked >= P->n_in_use) break;
                const auto now = clk::now();
                if (now >= t_long) break;
                if (now >= t_short && R->n_regular_queued + P->n_blocked >= P->n_prev_in_use) break;
                if (now < t_spin) {                    // poll: a futex wake-up costs tens of microseconds per arrival
                    const int seen = P->arrivals.load(std::memory_order_acquire);
                    lk.unlock();
                    while (P->arrivals.load(std::memory_order_acquire) == seen && clk::now() < t_spin) __builtin_ia32_pause();
                    pool_lock(lk);
                    continue;
                }
                ++P->n_cv_waiters; P->cv.wait_until(lk, now < t_short ? t_short : t_long); --P->n_cv_waiters;
            }
        }
    }
    // rounds execute in order: the previous round's leader holds exec_mu until its device call is through; this round stays open
    // (and keeps filling) while we wait for it
    std::lock_guard<std::mutex> ex(P->exec_mu);
    int k;
    {
        std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
        R->state = StreamPoolC::Round::FIRING;
        ++P->open_id;                                  // later submitters fill the other buffer
        P->cv.notify_all();
        while (R->copies_pending > 0) { ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters; }
        k = R->k;
        P->n_prev_in_use = 0;                          // (element-wise into vectors sized at creation: nothing here allocates)
        for (int i = 0; i < s->S; ++i) { P->prev_active[i] = R->active[i] && P->in_use[i]; P->n_prev_in_use += P->prev_active[i]; }
    }
    const auto t_call0 = std::chrono::steady_clock::now();
    // Whatever happens in the device call, the round reaches DONE with a return code: its followers block in pool_collect without
    // a time-out, and no C++ exception may cross the extern "C" boundary above us.  (R->active is not written while the round fires.)
    int rc; std::string call_err;
    try { rc = streams_call(s, nullptr, k, nullptr, R->active.data(), DPDF_HOST_PTRS, R->pin_in, R->pin_out); if (rc) call_err = dpdf_last_error(); }
    catch (const std::exception& e) { rc = DPDF_E_RUNTIME; try { call_err = std::string("exception in the pool's device call: ") + e.what(); } catch (...) {} }
    catch (...) { rc = DPDF_E_RUNTIME; }
    {
        std::lock_guard<std::mutex> lk(P->mu);
        const auto t_call1 = std::chrono::steady_clock::now();
        P->t_wait += std::chrono::duration<double>(t_call0 - t_lead0).count();
        P->t_call += std::chrono::duration<double>(t_call1 - t_call0).count();
        if (P->last_done != std::chrono::steady_clock::time_point{}) P->t_gap += std::chrono::duration<double>(t_call0 - P->last_done).count();
        P->last_done = t_call1;
        R->rc = rc; R->err.swap(call_err);
        R->state = StreamPoolC::Round::DONE;
        R->readers_left = R->n_queued;
        ++P->device_calls; ++P->rounds;
        P->cv.notify_all();
    }
}
// wait for the round the group rode in and copy its rows out (one acquisition of the lock on either side of the copies)
static int pool_collect(dpdf_streams* s, StreamPoolC* P, int n, const int* idx, const int* slots, float* const* out_rows) {
    const dpdf_dims& d = s->m->d;
    std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
    const long id = P->slot_round[slots[idx[0]]];
    if (id < 0) return set_err(DPDF_E_STATE, "slot %d has no request in flight", slots[idx[0]]);
    StreamPoolC::Round* R = &P->rd[id & 1];
    if (P->spin_s > 0 && !(R->id == id && R->state == StreamPoolC::Round::DONE)) {
        // poll for the round's completion (the leader's device call takes hundreds of microseconds; being woken through the
        // condition variable adds tens more on the way back to the caller, in front of its NEXT submission)
        const auto t_spin = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(P->spin_s));
        lk.unlock();
        while (!(R->id.load(std::memory_order_acquire) == id && R->state.load(std::memory_order_acquire) == StreamPoolC::Round::DONE) &&
               std::chrono::steady_clock::now() < t_spin) __builtin_ia32_pause();
        pool_lock(lk);
    }
    while (!(R->id == id && R->state == StreamPoolC::Round::DONE)) { ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters; }
    const int rc = R->rc;
    const std::string err = rc ? R->err : std::string();
    const size_t row = (size_t)R->k * d.hop;
    const float* base = R->pin_out;
    lk.unlock();
    if (!rc) for (int j = 0; j < n; ++j) memcpy(out_rows[idx[j]], base + (size_t)slots[idx[j]] * row, row * sizeof(float));
    pool_lock(lk);
    for (int j = 0; j < n; ++j) P->slot_round[slots[idx[j]]] = -1;
    R->readers_left -= n;
    if (R->readers_left == 0) {                         // last reader recycles the buffer for round id + 2
        R->state = StreamPoolC::Round::OPEN; R->id = id + 2; R->n_queued = 0; R->n_regular_queued = 0; R->k = 0; R->has_leader = false; R->rc = 0;
        std::fill(R->active.begin(), R->active.end(), 0);
        if (P->n_cv_waiters) P->cv.notify_all();
    }
    if (rc) return set_err(rc, "%s", err.c_str());
    return DPDF_OK;
}
// n requests of ONE host thread (n = 1: a StreamEnhancer-shaped object's process()): slots[i] gets ks[i] whole hops from in_rows[i],
// out_rows[i] takes ks[i] * hop samples.  Requests with the same hop count ride in the same round, together with whatever
// other threads submit in the window.  flags bit 0: do not wait for other submitters (the caller knows it is alone).
extern "C" int dpdf_streams_submit_many(dpdf_streams* s, int n, const int* slots, const float* const* in_rows, const int* ks,
                                        float* const* out_rows, int flags) {
    if (!s || !slots || !in_rows || !ks || !out_rows) return set_err(DPDF_E_INVALID, "null argument");
    if (n <= 0) return DPDF_OK;
    {
        std::vector<unsigned char> seen(s->S, 0);
        for (int i = 0; i < n; ++i) {
            if (slots[i] < 0 || slots[i] >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", slots[i], s->S);
            if (ks[i] <= 0 || !in_rows[i] || !out_rows[i]) return set_err(DPDF_E_INVALID, "bad request %d", i);
            if (!s->primed[slots[i]]) return set_err(DPDF_E_STATE, "stream %d not primed: call dpdf_streams_prime with its first hop", slots[i]);
            if (seen[slots[i]]) return set_err(DPDF_E_INVALID, "slot %d appears twice", slots[i]);
            seen[slots[i]] = 1;
        }
    }
    StreamPoolC* P = pool_of(s);
    std::vector<char> done(n, 0);
    std::vector<int> grp; grp.reserve(n);
    int first_rc = DPDF_OK; std::string first_err;
    for (int i0 = 0; i0 < n; ++i0) {
        if (done[i0]) continue;
        const int k = ks[i0];                          // one group (= one round) per distinct hop count, in order of appearance
        grp.clear();
        for (int i = i0; i < n; ++i) if (!done[i] && ks[i] == k) { grp.push_back(i); done[i] = 1; }
        long lid = -1;
        int rc = pool_join(s, P, (int)grp.size(), grp.data(), slots, in_rows, k, &lid);
        if (!rc) {
            if (lid >= 0) pool_lead(s, P, lid, (flags & 1) != 0);
            rc = pool_collect(s, P, (int)grp.size(), grp.data(), slots, out_rows);
        }
        if (rc && !first_rc) { first_rc = rc; first_err = dpdf_last_error(); }
    }
    if (first_rc) return set_err(first_rc, "%s", first_err.c_str());
    return DPDF_OK;
}
// the same for n requests of equal hop count whose rows lie one after the other: in_block / out_block [n][k_hops * hop]
extern "C" int dpdf_streams_submit_block(dpdf_streams* s, int n, const int* slots, const float* in_block, int k_hops, float* out_block, int flags) {
    if (!s || !slots || !in_block || !out_block || n < 0 || k_hops <= 0) return set_err(DPDF_E_INVALID, "bad argument");
    const size_t row = (size_t)k_hops * s->m->d.hop;
    std::vector<const float*> in(n); std::vector<float*> out(n); std::vector<int> ks(n, k_hops);
    for (int i = 0; i < n; ++i) { in[i] = in_block + (size_t)i * row; out[i] = out_block + (size_t)i * row; }
    return dpdf_streams_submit_many(s, n, slots, in.data(), ks.data(), out.data(), flags);
}
extern "C" int dpdf_streams_submit_wait(dpdf_streams* s, int slot, const float* pcm, int k_hops, float* out, int flags) {
    return dpdf_streams_submit_many(s, 1, &slot, &pcm, &k_hops, &out, flags);
}
