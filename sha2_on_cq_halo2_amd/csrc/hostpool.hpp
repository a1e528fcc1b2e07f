// A few persistent host threads per context for the host-side glue of a proof that is worth spreading: scalar
// multiplications of commitments (f by linearity), folding the bit-plane sums of a large MSM launch.  Creating a
// std::thread costs 20-40 us, several times the work items themselves; a parked worker wakes in a few microseconds.
// Work functions must not throw (nothing may unwind across the C ABI); with no worker threads to be had everything
// runs on the waiting thread.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace cq {

class HostPool {
 public:
  struct Job {
    std::function<void(size_t)> fn;
    size_t count = 0;
    std::atomic<size_t> next{0}, done{0};
  };
  using Ticket = std::shared_ptr<Job>;

  explicit HostPool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) {
      try {
        threads_.emplace_back([this]() { run(); });
      } catch (...) {
        break;
      }
    }
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;

  // fn(0) .. fn(count - 1), in any order, on the workers (and on whoever waits); returns at once
  Ticket submit(size_t count, std::function<void(size_t)> fn) {
    Ticket j = std::make_shared<Job>();
    j->fn = std::move(fn);
    j->count = count;
    if (count && !threads_.empty()) {
      {
        std::lock_guard<std::mutex> lk(mu_);
        queue_.push_back(j);
      }
      cv_.notify_all();
    }
    return j;
  }
  // helps with the job's remaining items, then waits for the ones in flight
  void wait(const Ticket& j) {
    if (!j) return;
    work_on(*j);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&]() { return j->done.load() >= j->count; });
  }
  void parallel_for(size_t count, std::function<void(size_t)> fn) { wait(submit(count, std::move(fn))); }

 private:
  void work_on(Job& j) {
    for (;;) {
      const size_t i = j.next.fetch_add(1);
      if (i >= j.count) return;
      j.fn(i);
      if (j.done.fetch_add(1) + 1 == j.count) {
        std::lock_guard<std::mutex> lk(mu_);  // pairs with the predicate check in wait()
        done_cv_.notify_all();
      }
    }
  }
  void run() {
    for (;;) {
      Ticket j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return stop_ || !queue_.empty(); });
        if (stop_) return;
        j = queue_.front();
        if (j->next.load() >= j->count) {  // exhausted: drop it and look again
          queue_.pop_front();
          continue;
        }
      }
      work_on(*j);
    }
  }
  std::vector<std::thread> threads_;
  std::deque<Ticket> queue_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  bool stop_ = false;
};

}  // namespace cq
