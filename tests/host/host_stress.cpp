// HIP-free unit over the host-side glue of the prover -- the worker pool (hostpool.hpp), the jump-ahead bulk form of the
// harness RNG (xoshiro.hpp), the transcript hash (blake2b.hpp) -- built by tests/test_sanitizers_cpu.py with
// -fsanitize=thread and with -fsanitize=address,undefined.  The threading patterns are the prover's own
// (csrc/prover.hip): a helper thread drawing the random polynomial chunk by chunk THROUGH the pool while the main thread
// submits jobs to the same pool and waits for them (f by linearity, MSM folds); batch lanes = several such main threads.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "blake2b.hpp"
#include "hostpool.hpp"
#include "xoshiro.hpp"

using namespace cq;

static int fails = 0;
#define EXPECT(c)                                             \
  do {                                                        \
    if (!(c)) {                                               \
      fails++;                                                \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
    }                                                         \
  } while (0)

static void seed_state(uint64_t seed, uint64_t st[4]) {  // splitmix64, as cq_xoshiro256ss_seed
  for (int i = 0; i < 4; i++) {
    seed += 0x9E3779B97F4A7C15ull;
    uint64_t z = seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    st[i] = z ^ (z >> 31);
  }
}

static void print_digest(const char* tag, const Blake2b& b) {
  uint8_t out[64];
  b.finalize_clone(out);
  printf("%s ", tag);
  for (int i = 0; i < 64; i++) printf("%02x", out[i]);
  printf("\n");
}

int main() {
  // ---- blake2b: digests for the test to compare with hashlib (person = "Halo2-Transcript", 64-byte output) ----
  {
    Blake2b b;
    b.init(64, "Halo2-Transcript");
    print_digest("blake2b_empty", b);
    std::vector<uint8_t> msg(1000);
    for (size_t i = 0; i < msg.size(); i++) msg[i] = (uint8_t)(i * 7 + 1);
    b.update(msg.data(), 1);
    print_digest("blake2b_1", b);  // finalize_clone leaves the state usable (squeeze_challenge, transcript.rs:214-219)
    b.update(msg.data() + 1, 127);
    print_digest("blake2b_128", b);
    b.update(msg.data() + 128, 1);
    print_digest("blake2b_129", b);
    b.update(msg.data() + 129, 871);
    print_digest("blake2b_1000", b);
  }
  // ---- xoshiro_fill: pool / fresh threads / serial give the same words and the same final state ----
  {
    HostPool pool(6);
    for (size_t count : {(size_t)0, (size_t)1, (size_t)70000, (size_t)((1u << 18) + 13), (size_t)1 << 20}) {
      uint64_t a[4], b[4], c[4];
      seed_state(count + 1, a);
      memcpy(b, a, sizeof a);
      memcpy(c, a, sizeof a);
      std::vector<uint64_t> wa(count), wb(count), wc(count);
      xoshiro_fill_serial(a, wa.data(), count);
      xoshiro_fill(b, wb.data(), count, 8, &pool);
      xoshiro_fill(c, wc.data(), count, 5, nullptr);
      EXPECT(wa == wb && wa == wc);
      EXPECT(memcmp(a, b, sizeof a) == 0 && memcmp(a, c, sizeof a) == 0);
    }
  }
  // ---- the prover's pattern, several lanes at once on one pool ----
  {
    HostPool pool(6);
    const int lanes = 3, rounds = 40;
    std::atomic<uint64_t> grand{0};
    std::vector<std::thread> mains;
    for (int lane = 0; lane < lanes; lane++) {
      mains.emplace_back([&, lane]() {
        for (int r = 0; r < rounds; r++) {
          uint64_t st[4], ref[4];
          seed_state(1000 * lane + r, st);
          memcpy(ref, st, sizeof st);
          const size_t words = (size_t)1 << 17;
          std::vector<uint64_t> drawn(words), expect(words);
          std::atomic<bool> done{false};
          std::thread drawer([&]() {  // RandomPolyDrawer: chunks through the pool from a helper thread
            for (size_t off = 0; off < words; off += words / 4) xoshiro_fill(st, drawn.data() + off, words / 4, 8, &pool);
            done.store(true);
          });
          // meanwhile the main thread: a few short jobs with results in its own frame (f commitments, folds)
          uint64_t sums[7] = {0};
          HostPool::Ticket t = pool.submit(7, [&](size_t i) {
            uint64_t acc = 0;
            for (uint64_t x = 0; x < 2000; x++) acc += x * (i + 1);
            sums[i] = acc;
          });
          pool.parallel_for(3, [&](size_t i) { grand.fetch_add(i + 1); });
          pool.wait(t);
          for (size_t i = 0; i < 7; i++) EXPECT(sums[i] == 1999ull * 2000 / 2 * (i + 1));
          while (!done.load()) std::this_thread::yield();
          drawer.join();
          xoshiro_fill_serial(ref, expect.data(), words);
          EXPECT(drawn == expect && memcmp(ref, st, sizeof st) == 0);
        }
      });
    }
    for (auto& m : mains) m.join();
    EXPECT(grand.load() == (uint64_t)lanes * rounds * 6);
  }
  // ---- a pool without workers runs everything on the waiting thread; empty jobs; destruction with work queued ----
  {
    HostPool none(0);
    int hits = 0;
    none.parallel_for(5, [&](size_t) { hits++; });
    EXPECT(hits == 5);
    none.parallel_for(0, [&](size_t) { hits++; });
    EXPECT(hits == 5);
    for (int i = 0; i < 50; i++) {
      HostPool p(4);
      std::atomic<int> n{0};
      HostPool::Ticket t = p.submit(64, [&](size_t) { n.fetch_add(1); });
      p.wait(t);  // (a job's frame must outlive it: the prover always waits -- PoolJoin in prover.hip)
      EXPECT(n.load() == 64);
    }
  }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
