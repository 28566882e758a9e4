// host_pool_check.cpp — the copy pool of the stream seam's ring route (sctagger_amd/csrc/sct_copy_pool.h) on the host, alone: the
// protocol sct_stream.hip.h runs on it — chunk k's way in, a "device" that takes a while, chunk k's way out whose last part hands the
// slot to chunk k + S — with random job lengths, pools of 0 .. 6 threads and a waiting thread that takes jobs itself.  Checked: every
// part runs exactly once; a slot is never written by its next user before its previous user's way out has finished; wait() returns
// only when the group is through; an abandoned chain (an error cuts the call short) leaves nobody waiting.  Built by
// tests/test_core_host.py plainly and by tests/test_sanitizers.py under ThreadSanitizer.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../sctagger_amd/csrc/sct_copy_pool.h"

struct Chunk;
struct Call {
    CopyPool *pool;
    int n_chunks, S;
    std::unique_ptr<Chunk[]> ch;
    std::vector<std::atomic<int>> slot_owner;   // which chunk may write the slot (-1: nobody)
    std::atomic<int> errors{0};
    explicit Call(int slots) : slot_owner(slots) {}
};
struct Chunk {
    Call *call = nullptr;
    int k = 0, in_parts = 0, out_parts = 0;
    CopyGroup in, out;
    Chunk *next_user = nullptr;
    std::atomic<int> in_ran{0}, out_ran{0};
    std::atomic<unsigned> in_mask{0}, out_mask{0};
    unsigned work = 0;
};

static void spin(unsigned n) {
    volatile unsigned x = 0;
    for (unsigned i = 0; i < n; i++) x += i;
}

static void job_in(void *arg, int part, int parts) {
    Chunk *c = (Chunk *)arg;
    Call *call = c->call;
    if (parts != c->in_parts) call->errors++;
    if (call->slot_owner[(size_t)(c->k % call->S)].load() != c->k) call->errors++;   // the slot must be ours by now
    spin(c->work * 37u % 4000u);
    if (c->in_mask.fetch_or(1u << part) & (1u << part)) call->errors++;              // a part run twice
    c->in_ran++;
}
static void job_out(void *arg, int part, int parts) {
    Chunk *c = (Chunk *)arg;
    Call *call = c->call;
    if (parts != c->out_parts) call->errors++;
    if (c->in_ran.load() != c->in_parts) call->errors++;                               // the way out before the way in was through
    spin(c->work * 53u % 3000u);
    if (c->out_mask.fetch_or(1u << part) & (1u << part)) call->errors++;
    c->out_ran++;
}
static void submit_in(Chunk *c) { c->call->pool->submit(&c->in, job_in, c, c->in_parts); }
static void handover(void *arg) {   // run by whoever finishes the last part of the previous user's way out
    Chunk *next = (Chunk *)arg;
    Call *call = next->call;
    int expect = next->k - call->S;
    if (!call->slot_owner[(size_t)(next->k % call->S)].compare_exchange_strong(expect, next->k)) call->errors++;
    submit_in(next);
    call->pool->release(&next->in);
}

static int run_call(CopyPool &pool, std::mt19937 &rng, int n_chunks, int S, int L, int stop_at) {
    Call call(S);
    call.pool = &pool;
    call.n_chunks = n_chunks;
    call.S = S;
    call.ch.reset(new Chunk[(size_t)n_chunks]);
    for (auto &o : call.slot_owner) o = -1;
    for (int k = 0; k < n_chunks; k++) {
        Chunk &c = call.ch[(size_t)k];
        c.call = &call;
        c.k = k;
        c.in_parts = 1 + (int)(rng() % 7);
        c.out_parts = 1 + (int)(rng() % 5);
        c.work = (unsigned)rng();
        if (k >= S) {
            call.ch[(size_t)(k - S)].next_user = &c;
            pool.hold(&c.in);
        }
    }
    std::vector<char> out_submitted((size_t)n_chunks, 0);
    int first_in = 0;
    for (int k = 0; k < S && k < n_chunks; k++) {
        call.slot_owner[(size_t)k] = k;
        submit_in(&call.ch[(size_t)k]);
        first_in = k + 1;
    }
    // the calling thread: wait for chunk k's way in, "queue it on the device", finish chunk k - L + 1
    const int last = stop_at >= 0 ? stop_at : n_chunks + L - 1;
    for (int k = 0; k < n_chunks + L - 1 && k < last; k++) {
        if (k < n_chunks) {
            pool.wait(&call.ch[(size_t)k].in);
            if (call.ch[(size_t)k].in_ran.load() != call.ch[(size_t)k].in_parts) call.errors++;
        }
        const int j = k - (L - 1);
        if (j >= 0) {
            Chunk &c = call.ch[(size_t)j];
            spin(c.work % 2000u);   // the device
            if (c.next_user) {
                c.out.then = handover;
                c.out.then_arg = c.next_user;
            }
            out_submitted[(size_t)j] = 1;
            pool.submit(&c.out, job_out, &c, c.out_parts);
        }
    }
    // quiesce, as StreamCall::ring_quiesce does: only what was queued, or will be by a way out that was
    for (int k = 0; k < n_chunks; k++) {
        Chunk &c = call.ch[(size_t)k];
        if (k < S && first_in > k) pool.wait(&c.in);
        if (out_submitted[(size_t)k]) {
            pool.wait(&c.out);
            if (c.out_ran.load() != c.out_parts) call.errors++;
            if (c.next_user) {
                pool.wait(&c.next_user->in);
                if (c.next_user->in_ran.load() != c.next_user->in_parts) call.errors++;
            }
        }
    }
    if (stop_at < 0)
        for (int k = 0; k < n_chunks; k++)
            if (call.ch[(size_t)k].in_ran != call.ch[(size_t)k].in_parts || call.ch[(size_t)k].out_ran != call.ch[(size_t)k].out_parts) call.errors++;
    return call.errors.load();
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(12345);
    int errors = 0, calls = 0;
    for (int threads : {0, 1, 2, 3, 6}) {
        CopyPool pool(threads);
        for (int r = 0; r < rounds; r++) {
            const int n_chunks = 1 + (int)(rng() % 14), L = 1 + (int)(rng() % 4), S = L + 1 + (int)(rng() % 3);
            const int stop_at = (rng() % 5 == 0) ? (int)(rng() % (unsigned)(n_chunks + L)) : -1;   // now and then an "error" cuts the call short
            errors += run_call(pool, rng, n_chunks, S < n_chunks ? S : n_chunks, L, stop_at);
            calls++;
        }
    }
    if (sct_effective_cpus() < 1) errors++;
    if (errors) {
        printf("FAILED: %d errors in %d calls\n", errors, calls);
        return 1;
    }
    printf("OK %d calls\n", calls);
    return 0;
}
