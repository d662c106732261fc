// copy_pool.hpp -- the copy threads of the host-memory entry points (gft_scan / gft_process from pageable caller memory).
// A few threads that live as long as the handle and split one memcpy between them: the staging of a batch is a dozen chunks
// for the link to stay busy, and starting twelve std::threads for every chunk cost as much as copying a small one (the
// reason the first chunk of an upload could not be small).  copy() is called by one thread at a time (the handle's mutex).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace gft {

class CopyPool {
public:
    explicit CopyPool(unsigned workers) {
        th_.reserve(workers);
        try {
            for (unsigned t = 0; t < workers; t++) th_.emplace_back([this, t] { work(t + 1); });
        } catch (...) {
            // (a thread that could not be started: the pool works with the ones that were)
        }
    }
    CopyPool(const CopyPool&) = delete;
    CopyPool& operator=(const CopyPool&) = delete;
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : th_)
            if (t.joinable()) t.join();
    }
    unsigned threads() const { return (unsigned)th_.size() + 1; }

    // dst[0, n) = src[0, n), split evenly over the workers and the caller; returns when every byte is there
    void copy(void* dst, const void* src, size_t n) noexcept {
        const unsigned parts = threads();
        const size_t part = (n + parts - 1) / parts;
        if (n < (1u << 20) || th_.empty()) { memcpy(dst, src, n); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = (uint8_t*)dst; src_ = (const uint8_t*)src; n_ = n; part_ = part;
            pending_ = (unsigned)th_.size();
            gen_++;
        }
        cv_work_.notify_all();
        memcpy(dst, src, part < n ? part : n);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    void work(unsigned t) noexcept {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            uint8_t* d = dst_;
            const uint8_t* s = src_;
            const size_t n = n_, part = part_;
            lk.unlock();
            const size_t a = t * part < n ? t * part : n, b = (t + 1) * part < n ? (t + 1) * part : n;
            if (b > a) memcpy(d + a, s + a, b - a);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    uint64_t gen_ = 0;
    unsigned pending_ = 0;
    bool stop_ = false;
    uint8_t* dst_ = nullptr;
    const uint8_t* src_ = nullptr;
    size_t n_ = 0, part_ = 0;
};

}  // namespace gft
