// crf_pool.h -- a small persistent thread pool with a spin-then-sleep hand-off, shared by the device group (one worker
// per device, group.cpp) and the host-output path (copier threads, api.cpp).  Internal to libcorrfield.so.
//
// Why not a plain condition variable: evaluations of an interactive session follow each other within milliseconds, the
// per-device work of an 8-GPU evaluation is ~0.1 ms, and a futex wake-up out of an idle state costs 30-80 us each way.
// So publication is lock-free (a job pointer, then a release-increment of a generation counter), idle workers first SPIN
// on the generation for a bounded time (CRF_POOL_SPIN_US, default 2000 us after their last job) and only then sleep on a
// condition variable; the caller wakes sleepers only when the sleeper count says there are any.  Completion is a
// counter the caller spins on the same way.  In-job rendezvous (barrier) is a spinning sense barrier: all workers are
// awake while a job runs.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <immintrin.h>

namespace crf {

class SpinPool {
public:
    // init(r) runs once on worker r before its first job (e.g. hipSetDevice)
    // spin_seconds: how long an idle worker (and a waiting caller) polls before it sleeps
    explicit SpinPool(int n, std::function<void(int)> init = nullptr, double spin_seconds = 2000e-6)
        : n_(n), spin_seconds_(spin_seconds), status_(size_t(n), 0) {
        if (const char* e = getenv("CRF_POOL_SPIN_US"); e && atof(e) >= 0.0) spin_seconds_ = atof(e) * 1e-6;
        for (int r = 0; r < n; r++)
            threads_.emplace_back([this, r, init] {
                if (init) init(r);
                loop(r);
            });
    }
    ~SpinPool() {
        wait();
        stop_.store(true, std::memory_order_seq_cst);
        publish();
        for (auto& t : threads_) t.join();
    }
    SpinPool(const SpinPool&) = delete;
    SpinPool& operator=(const SpinPool&) = delete;
    int size() const { return n_; }

    // Hands `job` to every worker (job(r) on worker r) and returns at once; the job object must stay alive until wait().
    void start(const std::function<int(int)>& job) {
        wait();
        job_ = &job;
        remaining_.store(n_, std::memory_order_relaxed);
        running_ = true;
        publish();
    }
    // Returns when every worker has finished the started job: the first non-zero status (by worker index), or 0.
    int wait() {
        if (!running_) return 0;
        if (!spin_until([this] { return remaining_.load(std::memory_order_acquire) == 0; }, spin_seconds_)) {
            std::unique_lock<std::mutex> lk(m_);
            caller_sleeping_.store(true, std::memory_order_seq_cst);
            done_cv_.wait(lk, [this] { return remaining_.load(std::memory_order_seq_cst) == 0; });
            caller_sleeping_.store(false, std::memory_order_relaxed);
        }
        running_ = false;
        job_ = nullptr;
        for (int s : status_)
            if (s) return s;
        return 0;
    }
    int run(const std::function<int(int)>& job) {
        start(job);
        return wait();
    }
    // rendezvous of all workers inside a job (every worker must call it the same number of times)
    void barrier() {
        const unsigned phase = barrier_phase_.load(std::memory_order_acquire);
        if (barrier_count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            barrier_count_.store(0, std::memory_order_relaxed);
            barrier_phase_.store(phase + 1, std::memory_order_release);
        } else {
            for (unsigned i = 0; barrier_phase_.load(std::memory_order_acquire) == phase; i++) {
                if ((i & 1023u) == 1023u) std::this_thread::yield();
                else _mm_pause();
            }
        }
    }

private:
    template <class Pred>
    static bool spin_until(Pred done, double seconds) {
        if (done()) return true;
        if (seconds <= 0.0) return false;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned i = 1;; i++) {
            if (done()) return true;
            if ((i & 255u) == 0u) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
                std::this_thread::yield();  // let a thread that shares the core (oversubscribed hosts) make progress
            } else {
                _mm_pause();
            }
        }
    }
    void publish() {
        generation_.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
    }
    void loop(int r) {
        unsigned long seen = 0;
        for (;;) {
            auto changed = [&] { return generation_.load(std::memory_order_seq_cst) != seen; };
            if (!spin_until(changed, spin_seconds_)) {
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1, std::memory_order_seq_cst);
                cv_.wait(lk, changed);
                sleepers_.fetch_sub(1, std::memory_order_seq_cst);
            }
            seen = generation_.load(std::memory_order_seq_cst);
            if (stop_.load(std::memory_order_seq_cst)) return;
            const std::function<int(int)>* job = job_;  // published before the generation increment
            status_[size_t(r)] = (*job)(r);
            if (remaining_.fetch_sub(1, std::memory_order_seq_cst) == 1 && caller_sleeping_.load(std::memory_order_seq_cst)) {
                std::lock_guard<std::mutex> lk(m_);
                done_cv_.notify_all();
            }
        }
    }
    int n_;
    double spin_seconds_;
    std::vector<std::thread> threads_;
    std::vector<int> status_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<int(int)>* job_ = nullptr;
    bool running_ = false;
    std::atomic<unsigned long> generation_{0};
    std::atomic<int> remaining_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> caller_sleeping_{false};
    std::atomic<bool> stop_{false};
    std::atomic<int> barrier_count_{0};
    std::atomic<unsigned> barrier_phase_{0};
};

}  // namespace crf
