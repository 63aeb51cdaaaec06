// rtx_bvh.h — host builder of the 16-wide box-list hierarchy (see rtx_bvh.cpp).
#pragma once
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "rtx_device.h"

// The host-side builds' worker threads (octree subtrees, SAH passes and subtrees, record fills): created once per rtx_scene_create
// and reused for every parallel region (creating threads per region cost more than the regions of the SAH build's top levels).
// run(n, fn) calls fn(0 .. n-1), tasks handed out dynamically, the caller's thread working too; it returns when all are done.
class RtxPool {
public:
    explicit RtxPool(unsigned n_threads) {
        for (unsigned i = 1; i < n_threads; ++i) workers_.emplace_back([this]() { loop(); });
    }
    ~RtxPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    unsigned size() const { return (unsigned)workers_.size() + 1u; }
    void run(size_t n_tasks, const std::function<void(size_t)> &fn) {
        if (n_tasks == 0) return;
        if (workers_.empty() || n_tasks == 1) {
            for (size_t t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_tasks_ = n_tasks;
            next_.store(0);
            pending_ = workers_.size();
            ++epoch_;
        }
        cv_.notify_all();
        for (size_t t; (t = next_.fetch_add(1)) < n_tasks;) fn(t);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop() {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *fn;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                fn = fn_;
                n = n_tasks_;
            }
            for (size_t t; (t = next_.fetch_add(1)) < n;) (*fn)(t);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_tasks_ = 0, pending_ = 0, epoch_ = 0;
    std::atomic<size_t> next_{0};
    bool stop_ = false;
};

struct RtxPrimBox {
    double mn[3], mx[3];
    uint32_t ref;        // RTX_WREF_TRI | triangle index  or  RTX_WREF_SPHERE | sphere index
};
struct RtxWideStats {
    int64_t nodes = 0, prim_children = 0, nodes4 = 0;
    int depth = 0, depth4 = 0;
};
// One binned-SAH binary tree over the prims, cut twice: into 16-wide box lists (`out`, the cooperative shadow walk) and into 4-wide
// nodes (`out4`, the per-lane closest-hit and any-hit walks).  Boxes are stored relative to `centre`, rounded outward to f32.
// pool: worker threads of the build (the trees do not depend on their number).
void rtx_build_hierarchies(const std::vector<RtxPrimBox> &prims, const double centre[3], std::vector<DWideChild> &out,
                           std::vector<DNode4> &out4, RtxWideStats *stats, RtxPool &pool);
