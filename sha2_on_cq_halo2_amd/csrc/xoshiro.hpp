// Host-only: the harness generator xoshiro256** with jump-ahead, so that one long run of draws can be produced by
// several threads and still be THE stream (word i is the i-th output after the given state, whoever computes it).
//
// The state transition is linear over GF(2) on the 256 state bits.  T^(2^j) are kept as 256 x 256 bit matrices
// (column form: col[b] = image of basis vector b; built by repeated squaring, ~50 us each, cached for the process);
// a jump by `steps` applies T^(2^j) for every set bit j of `steps` -- a few hundred word XORs per bit.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "hostpool.hpp"

namespace cq {

struct XoshiroState {
  uint64_t s[4];
};

static inline void xoshiro_step(uint64_t* s) {
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = (s[3] << 45) | (s[3] >> 19);
}

// `count` outputs into dst, advancing the state (the generator of cq_xoshiro256ss_next_u64)
static inline void xoshiro_fill_serial(uint64_t* st, uint64_t* dst, size_t count) {
  uint64_t s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3];
  for (size_t i = 0; i < count; i++) {
    const uint64_t r5 = s1 * 5;
    dst[i] = ((r5 << 7) | (r5 >> 57)) * 9;
    const uint64_t t = s1 << 17;
    s2 ^= s0;
    s3 ^= s1;
    s1 ^= s2;
    s0 ^= s3;
    s2 ^= t;
    s3 = (s3 << 45) | (s3 >> 19);
  }
  st[0] = s0; st[1] = s1; st[2] = s2; st[3] = s3;
}

struct XoshiroJump {
  struct Mat {
    XoshiroState col[256];
  };
  static void apply(const Mat& m, uint64_t* s) {  // s = m * s
    uint64_t r[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; w++) {
      uint64_t bits = s[w];
      while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        const XoshiroState& c = m.col[w * 64 + b];
        r[0] ^= c.s[0]; r[1] ^= c.s[1]; r[2] ^= c.s[2]; r[3] ^= c.s[3];
      }
    }
    s[0] = r[0]; s[1] = r[1]; s[2] = r[2]; s[3] = r[3];
  }
  // T^(2^j), built on demand
  static const Mat& pow2(unsigned j) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<Mat>> cache;  // (owned: nothing is left behind for a leak checker at exit)
    std::lock_guard<std::mutex> lock(mu);
    while (cache.size() <= j) {
      std::unique_ptr<Mat> m(new Mat);
      if (cache.empty()) {
        for (int b = 0; b < 256; b++) {
          uint64_t e[4] = {0, 0, 0, 0};
          e[b >> 6] = 1ull << (b & 63);
          xoshiro_step(e);
          for (int w = 0; w < 4; w++) m->col[b].s[w] = e[w];
        }
      } else {
        const Mat& p = *cache.back();
        for (int b = 0; b < 256; b++) {  // (p * p) e_b = p * (p e_b)
          uint64_t v[4] = {p.col[b].s[0], p.col[b].s[1], p.col[b].s[2], p.col[b].s[3]};
          apply(p, v);
          for (int w = 0; w < 4; w++) m->col[b].s[w] = v[w];
        }
      }
      cache.push_back(std::move(m));
    }
    return *cache[j];
  }
  static void jump(uint64_t* s, uint64_t steps) {
    for (unsigned j = 0; steps; j++, steps >>= 1)
      if (steps & 1) apply(pow2(j), s);
  }
};

// the same words and the same final state as xoshiro_fill_serial, produced by up to `threads` threads: the context's
// parked workers when `pool` is given (no thread is created: ~30 us each, more than a part's work), fresh ones otherwise
static inline void xoshiro_fill(uint64_t* st, uint64_t* dst, size_t count, unsigned threads, HostPool* pool = nullptr) {
  const size_t MIN_PER_THREAD = (size_t)1 << 15;
  if (threads > count / MIN_PER_THREAD) threads = (unsigned)(count / MIN_PER_THREAD);
  if (threads <= 1) {
    xoshiro_fill_serial(st, dst, count);
    return;
  }
  const size_t per = count / threads;
  std::vector<XoshiroState> start(threads);
  for (int w = 0; w < 4; w++) start[0].s[w] = st[w];
  for (unsigned t = 1; t < threads; t++) {
    start[t] = start[t - 1];
    XoshiroJump::jump(start[t].s, per);
  }
  auto part = [&start, dst, per, count, threads](size_t t) {
    const size_t off = t * per, cnt = (t + 1 == threads) ? count - off : per;
    xoshiro_fill_serial(start[t].s, dst + off, cnt);
  };
  if (pool) {
    pool->parallel_for(threads, part);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++) {
      try {
        th.emplace_back(part, (size_t)t);
      } catch (...) {  // no thread to be had: this part on the calling thread
        part(t);
      }
    }
    for (auto& x : th) x.join();
  }
  for (int w = 0; w < 4; w++) st[w] = start[threads - 1].s[w];  // the last part ends where the whole run ends
}

}  // namespace cq
