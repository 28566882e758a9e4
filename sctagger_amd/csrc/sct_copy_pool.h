// sct_copy_pool.h — a small pool of host threads that move caller memory into and out of the context's page-locked ring
// (sct_stream.hip.h), and the count of CPUs a pool may be sized by.  Host code only.
//
// The reference hands its seam ordinary (pageable) arrays (scTagger.py:616-627: RawArray); the device can neither read them in
// place nor copy them without the runtime staging every piece through its own bounce buffer on the calling thread.  This pool is
// the library's own staging: several threads memcpy chunk k+1 into page-locked memory while the kernels read chunk k there.
#pragma once
#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

// CPUs this process may really use: its affinity mask cut down to its cgroup's CPU quota (v2 cpu.max, v1 cfs files); SCT_CPUS
// overrides that count; with LOCAL_WORLD_SIZE ranks on the node, a rank's share of them.  (A container on a 256-thread host may own 16 of them; a pool sized by the 256 spends the quota in the first
// milliseconds of every scheduler period and is then suspended — sctagger_amd/streamcli.py effective_cpus() is the same rule.)
static inline int sct_effective_cpus() {
    int cores = 0;
    if (const char *env = getenv("SCT_CPUS")) cores = atoi(env);
    if (cores <= 0) {
        cpu_set_t set;
        cores = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)sysconf(_SC_NPROCESSORS_ONLN);
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32];
            double p = 0;
            if (fscanf(f, "%31s %lf", q, &p) == 2 && q[0] != 'm' && p > 0) quota = atof(q) / p;
            fclose(f);
        } else {
            FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            double q = 0, p = 0;
            if (fq && fp && fscanf(fq, "%lf", &q) == 1 && fscanf(fp, "%lf", &p) == 1 && q > 0 && p > 0) quota = q / p;
            if (fq) fclose(fq);
            if (fp) fclose(fp);
        }
        if (quota > 0) {
            const int by_quota = (int)(quota + 0.5) < 1 ? 1 : (int)(quota + 0.5);
            if (by_quota < cores) cores = by_quota;
        }
    }
    // one process per GPU (torchrun / shard.launch_ranks set LOCAL_WORLD_SIZE): a rank takes its share
    if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = atoi(lw);
        if (ranks > 1) cores /= ranks;
    }
    return cores < 1 ? 1 : cores;
}

// Jobs that belong together (the parts of one chunk's copy): `remaining` counts the parts not yet finished — and, while a group
// waits to be queued by another group's last part (`then`), one more: CopyPool::hold / release.
struct CopyGroup {
    std::atomic<int> remaining{0};
    void (*then)(void *) = nullptr;   // run once by the thread that finishes the group's last part (set before the group is queued)
    void *then_arg = nullptr;
    bool done() const { return remaining.load(std::memory_order_acquire) == 0; }
};

class CopyPool {
  public:
    typedef void (*Fn)(void *arg, int part, int parts);

    explicit CopyPool(int threads) {
        if (threads < 0) threads = 0;
        th_.reserve((size_t)threads);
        for (int i = 0; i < threads; i++) th_.emplace_back([this] { work(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        work_cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    int threads() const { return (int)th_.size(); }

    // queue fn(arg, p, parts) for p = 0 .. parts-1
    void submit(CopyGroup *g, Fn fn, void *arg, int parts) {
        if (parts <= 0) return;
        g->remaining.fetch_add(parts, std::memory_order_acq_rel);
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int p = 0; p < parts; p++) q_.push_back(Job{fn, arg, p, parts, g});
        }
        if (parts == 1) work_cv_.notify_one();
        else work_cv_.notify_all();
        done_cv_.notify_all();   // (a thread in wait() takes jobs too)
    }

    // A group that somebody else will queue later (a `then` of another group): held open until then, so that wait() on it does
    // not return early.  release() after the submit().
    void hold(CopyGroup *g) { g->remaining.fetch_add(1, std::memory_order_acq_rel); }
    void release(CopyGroup *g) { finished_one(g); }

    // until every part of g has finished; the calling thread takes queued jobs meanwhile (a pool of no threads still works)
    void wait(CopyGroup *g) {
        std::unique_lock<std::mutex> lk(mu_);
        while (!g->done()) {
            if (!q_.empty()) {
                Job j = q_.front();
                q_.pop_front();
                lk.unlock();
                run(j);
                lk.lock();
            } else {
                done_cv_.wait(lk);
            }
        }
    }

  private:
    struct Job {
        Fn fn;
        void *arg;
        int part, parts;
        CopyGroup *g;
    };
    void finished_one(CopyGroup *g) {
        // (read while this part still keeps the group open: once `remaining` is 0 a waiter may be through and the group gone)
        void (*const then)(void *) = g->then;
        void *const then_arg = g->then_arg;
        if (g->remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            // the group's last part.  Its `then` queues what was waiting for it — a group that is held open (hold), so whoever waits
            // for both cannot be through before `then` has run
            if (then) then(then_arg);
            std::lock_guard<std::mutex> lk(mu_);   // (the waiter checks done() under this lock: no lost wake-up)
            done_cv_.notify_all();
        }
    }
    void run(const Job &j) {
        j.fn(j.arg, j.part, j.parts);
        finished_one(j.g);
    }
    void work() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            while (!stop_ && q_.empty()) work_cv_.wait(lk);
            if (q_.empty()) return;   // stop_
            Job j = q_.front();
            q_.pop_front();
            lk.unlock();
            run(j);
            lk.lock();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable work_cv_, done_cv_;
    std::deque<Job> q_;
    bool stop_ = false;
};
