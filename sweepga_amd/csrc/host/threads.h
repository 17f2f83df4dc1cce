// run(threads, body): body(0) .. body(threads - 1), each on its own host thread (body(0) on the caller's).
// Nothing escapes a worker: an exception inside a body is caught there and re-thrown (as std::bad_alloc) on the calling
// thread after every worker has been joined; a worker that cannot be started (std::system_error from std::thread) has its
// body run on the calling thread instead.  So a caller behind the C ABI maps one catch to SWG_ERR_OOM and no path reaches
// std::terminate.
#ifndef SWG_HOST_THREADS_H
#define SWG_HOST_THREADS_H

#include <atomic>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

namespace swg_host {

template <class F>
inline void run(int threads, F&& body) {
  std::atomic<bool> failed{false};
  auto guarded = [&](int t) {
    try {
      body(t);
    } catch (...) {
      failed.store(true, std::memory_order_relaxed);
    }
  };
  if (threads <= 1) {
    guarded(0);
  } else {
    std::vector<std::thread> pool;
    std::vector<int> inline_ts;
    try {
      pool.reserve((size_t)threads - 1);
    } catch (...) {
      for (int t = 0; t < threads; ++t) guarded(t);
      if (failed.load()) throw std::bad_alloc();
      return;
    }
    for (int t = 1; t < threads; ++t) {
      try {
        pool.emplace_back([&guarded, t] { guarded(t); });
      } catch (const std::system_error&) {
        guarded(t);  // no thread to be had: do its share here
      }
    }
    guarded(0);
    for (auto& th : pool) th.join();
  }
  if (failed.load()) throw std::bad_alloc();
}

}  // namespace swg_host
#endif
