// BLAKE2b (RFC 7693) with personalisation -- the Fiat-Shamir hash of the reference's transcript
// (halo2_proofs/src/transcript.rs:179-184: hash_length 64, personal "Halo2-Transcript").
// The reference takes it from the `blake2b_simd` crate (not vendored); this is a host-only
// restatement of the RFC.  The state is copyable: `squeeze_challenge` finalises a clone (:214-219).
#pragma once
#include <stdint.h>
#include <string.h>

namespace cq {

struct Blake2b {
  uint64_t h[8];
  uint64_t t[2];
  uint8_t buf[128];
  size_t buflen;

  static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

  void init(uint8_t outlen, const char personal[16]) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint8_t p[64];
    memset(p, 0, 64);
    p[0] = outlen;  // digest length
    p[1] = 0;       // key length
    p[2] = 1;       // fanout
    p[3] = 1;       // depth
    memcpy(p + 48, personal, 16);
    for (int i = 0; i < 8; i++) {
      uint64_t w;
      memcpy(&w, p + 8 * i, 8);
      h[i] = IV[i] ^ w;
    }
    t[0] = t[1] = 0;
    buflen = 0;
  }

  void compress(const uint8_t block[128], bool last) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    static const uint8_t S[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) memcpy(&m[i], block + 8 * i, 8);
    for (int i = 0; i < 8; i++) {
      v[i] = h[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t[0];
    v[13] ^= t[1];
    if (last) v[14] = ~v[14];
#define CQ_B2G(a, b, c, d, x, y)      \
  v[a] = v[a] + v[b] + (x);           \
  v[d] = rotr(v[d] ^ v[a], 32);       \
  v[c] = v[c] + v[d];                 \
  v[b] = rotr(v[b] ^ v[c], 24);       \
  v[a] = v[a] + v[b] + (y);           \
  v[d] = rotr(v[d] ^ v[a], 16);       \
  v[c] = v[c] + v[d];                 \
  v[b] = rotr(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
      const uint8_t* s = S[r];
      CQ_B2G(0, 4, 8, 12, m[s[0]], m[s[1]])
      CQ_B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
      CQ_B2G(2, 6, 10, 14, m[s[4]], m[s[5]])
      CQ_B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
      CQ_B2G(0, 5, 10, 15, m[s[8]], m[s[9]])
      CQ_B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
      CQ_B2G(2, 7, 8, 13, m[s[12]], m[s[13]])
      CQ_B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef CQ_B2G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }

  void update(const uint8_t* in, size_t len) {
    while (len) {
      if (buflen == 128) {
        t[0] += 128;
        if (t[0] < 128) t[1]++;
        compress(buf, false);
        buflen = 0;
      }
      size_t take = 128 - buflen;
      if (take > len) take = len;
      memcpy(buf + buflen, in, take);
      buflen += take;
      in += take;
      len -= take;
    }
  }

  // finalises a COPY of the state; *this stays usable
  void finalize_clone(uint8_t out[64]) const {
    Blake2b c = *this;
    c.t[0] += c.buflen;
    if (c.t[0] < c.buflen) c.t[1]++;
    memset(c.buf + c.buflen, 0, 128 - c.buflen);
    c.compress(c.buf, true);
    memcpy(out, c.h, 64);
  }
};

}  // namespace cq
