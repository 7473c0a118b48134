// Marlin's Fiat–Shamir RNG on the host side of the library: the product counterpart of
// /root/reference/marlin/src/fs_rng.rs:11-70 (`FiatShamirRng`: merlin transcript "MARLINSEED" -> 32-byte seed ->
// ChaChaRng) and of the verifier-message sampling the prover replays (marlin/src/lib.rs:105-158,
// ahp/verifier.rs:41-87,118-127).
//
// Third-party pieces the reference takes from crates (merlin 2.0 -> STROBE-128 over Keccak-f[1600]; rand_chacha 0.2
// `ChaChaRng` = ChaCha20 with a 64-bit block counter, 4 blocks buffered by rand_core 0.5's BlockRng; ark-ff 0.2
// `UniformRand for Fp256`) are implemented here from their specifications.  Pure host code: a few KiB of hashing per
// proof, nothing for the GPU to do.
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/zkp_accel.h"

namespace {

// ------------------------------------------------------------------------------------------------ Keccak-f[1600]
constexpr uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
constexpr int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void keccak_f1600(uint8_t st[200]) {
  uint64_t a[25];
  for (int i = 0; i < 25; i++) memcpy(&a[i], st + 8 * i, 8);           // little-endian host (x86-64)
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KECCAK_ROT[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= KECCAK_RC[rnd];
  }
  for (int i = 0; i < 25; i++) memcpy(st + 8 * i, &a[i], 8);
}

// ------------------------------------------------------------------------------------------------ STROBE-128 / merlin
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;
constexpr int STROBE_R = 166;

struct Strobe128 {
  uint8_t st[200];
  uint8_t pos = 0, pos_begin = 0, cur_flags = 0;
  explicit Strobe128(const uint8_t* label, size_t n) {
    memset(st, 0, sizeof st);
    const uint8_t head[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(st, head, 6);
    memcpy(st + 6, "STROBEv1.0.2", 12);
    keccak_f1600(st);
    meta_ad(label, n, false);
  }
  void run_f() {
    st[pos] ^= pos_begin;
    st[pos + 1] ^= 0x04;
    st[STROBE_R + 1] ^= 0x80;
    keccak_f1600(st);
    pos = pos_begin = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      st[pos++] ^= d[i];
      if (pos == STROBE_R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      d[i] = st[pos];
      st[pos++] = 0;
      if (pos == STROBE_R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;                                   // continuation of the current operation (same flags)
    const uint8_t old_begin = pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos != 0) run_f();
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(d, n);
  }
  void ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_A, more);
    absorb(d, n);
  }
  void prf(uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, more);
    squeeze(d, n);
  }
};

struct MerlinTranscript {
  Strobe128 s;
  MerlinTranscript(const uint8_t* label, size_t n) : s(reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11) {
    append_message(reinterpret_cast<const uint8_t*>("dom-sep"), 7, label, n);
  }
  static void le32(uint32_t v, uint8_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = (uint8_t)(v >> (8 * i));
  }
  void append_message(const uint8_t* label, size_t ln, const uint8_t* msg, size_t mn) {
    uint8_t len[4];
    le32((uint32_t)mn, len);
    s.meta_ad(label, ln, false);
    s.meta_ad(len, 4, true);
    s.ad(msg, mn, false);
  }
  void challenge_bytes(const uint8_t* label, size_t ln, uint8_t* out, size_t n) {
    uint8_t len[4];
    le32((uint32_t)n, len);
    s.meta_ad(label, ln, false);
    s.meta_ad(len, 4, true);
    s.prf(out, n, false);
  }
};

// ------------------------------------------------------------------------------------------------ ChaCha20 RNG
inline uint32_t rol32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5],
                     key[6],      key[7],      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
  uint32_t x[16];
  memcpy(x, in, sizeof x);
#define ZKP_QR(a, b, c, d)                                                      \
  x[a] += x[b]; x[d] = rol32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rol32(x[b] ^ x[c], 12); \
  x[a] += x[b]; x[d] = rol32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rol32(x[b] ^ x[c], 7)
  for (int i = 0; i < 10; i++) {
    ZKP_QR(0, 4, 8, 12); ZKP_QR(1, 5, 9, 13); ZKP_QR(2, 6, 10, 14); ZKP_QR(3, 7, 11, 15);
    ZKP_QR(0, 5, 10, 15); ZKP_QR(1, 6, 11, 12); ZKP_QR(2, 7, 8, 13); ZKP_QR(3, 4, 9, 14);
  }
#undef ZKP_QR
  for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

struct ChaChaRng {                                   // rand_chacha 0.2 ChaChaRng over rand_core 0.5 BlockRng
  static constexpr int BUF = 64;                     // 4 blocks
  uint32_t key[8];
  uint64_t counter = 0;
  uint32_t results[BUF];
  int index = BUF;
  void from_seed(const uint8_t seed[32]) {
    memcpy(key, seed, 32);
    counter = 0;
    index = BUF;
  }
  void generate_and_set(int idx) {
    for (int k = 0; k < 4; k++) chacha20_block(key, counter + k, results + 16 * k);
    counter += 4;
    index = idx;
  }
  uint64_t next_u64() {
    const int i = index;
    if (i < BUF - 1) {
      index += 2;
      return ((uint64_t)results[i + 1] << 32) | results[i];
    }
    if (i >= BUF) {
      generate_and_set(2);
      return ((uint64_t)results[1] << 32) | results[0];
    }
    const uint64_t x = results[BUF - 1];
    generate_and_set(1);
    return ((uint64_t)results[0] << 32) | x;
  }
};

// ------------------------------------------------------------------------------------------------ Fr on the host
#include <stdint.h>
namespace consts {
#include "field_constants.inc"
}
struct FrDesc {
  const uint32_t* mod;
  const uint32_t* one;
  uint32_t inv;
  int bits;
};
FrDesc fr_desc(int curve) {
  if (curve == ZKP_BN254) return {consts::Bn254Fr::MOD, consts::Bn254Fr::ONE, consts::Bn254Fr::INV, 254};
  return {consts::Bls381Fr::MOD, consts::Bls381Fr::ONE, consts::Bls381Fr::INV, 255};
}
bool geq(const uint32_t a[8], const uint32_t* m) {
  for (int i = 7; i >= 0; i--)
    if (a[i] != m[i]) return a[i] > m[i];
  return true;
}
// Montgomery product (CIOS, 32-bit limbs, R = 2^256), inputs < p
void mont_mul(const FrDesc& f, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  uint32_t t[10] = {0};
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 8; j++) {
      c += (uint64_t)a[j] * b[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * f.inv;
    c = (uint64_t)m * f.mod[0] + t[0];
    c >>= 32;
    for (int j = 1; j < 8; j++) {
      c += (uint64_t)m * f.mod[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
  if (t[8] || geq(t, f.mod)) {
    uint64_t br = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)t[i] - f.mod[i] - br;
      t[i] = (uint32_t)d;
      br = (d >> 63) & 1;
    }
  }
  memcpy(out, t, 32);
}

}  // namespace

struct zkp_fs_rng {
  uint8_t seed[32];
  ChaChaRng r;
  std::vector<uint8_t> buf;
  void reseed(const uint8_t* material, size_t n, bool chain) {
    buf.assign(material, material + n);
    if (chain) buf.insert(buf.end(), seed, seed + 32);       // absorb: H(new material || previous seed), fs_rng.rs:57-63
    MerlinTranscript t(reinterpret_cast<const uint8_t*>("MARLINSEED"), 10);
    t.append_message(reinterpret_cast<const uint8_t*>("Seed"), 4, buf.data(), buf.size());
    t.challenge_bytes(reinterpret_cast<const uint8_t*>("x"), 1, seed, 32);
    r.from_seed(seed);
  }
  // ark-ff 0.2 `Fp256::rand`: four u64 limbs, top limb masked to the modulus width, rejected when >= p; the accepted
  // integer is the element's MONTGOMERY representation (so it is handed out as-is in the ABI's layout)
  void rand_fr(int curve, uint32_t out[8]) {
    const FrDesc f = fr_desc(curve);
    for (;;) {
      uint64_t l[4];
      for (int i = 0; i < 4; i++) l[i] = r.next_u64();
      l[3] &= ~0ull >> (256 - f.bits);
      memcpy(out, l, 32);
      if (!geq(out, f.mod)) return;
    }
  }
};

extern "C" {

int32_t zkp_fs_rng_new(const uint8_t* seed_material, size_t len, zkp_fs_rng** out) {
  if (!out || (len && !seed_material)) return ZKP_ERR_BAD_ARG;
  zkp_fs_rng* g = new (std::nothrow) zkp_fs_rng();
  if (!g) return ZKP_ERR_OOM;
  g->reseed(seed_material, len, false);
  *out = g;
  return ZKP_OK;
}
int32_t zkp_fs_rng_free(zkp_fs_rng* g) {
  delete g;
  return ZKP_OK;
}
int32_t zkp_fs_rng_absorb(zkp_fs_rng* g, const uint8_t* material, size_t len) {
  if (!g || (len && !material)) return ZKP_ERR_BAD_ARG;
  g->reseed(material, len, true);
  return ZKP_OK;
}
int32_t zkp_fs_rng_seed(const zkp_fs_rng* g, uint8_t out32[32]) {
  if (!g || !out32) return ZKP_ERR_BAD_ARG;
  memcpy(out32, g->seed, 32);
  return ZKP_OK;
}
int32_t zkp_fs_rng_next_u64(zkp_fs_rng* g, uint64_t* out) {
  if (!g || !out) return ZKP_ERR_BAD_ARG;
  *out = g->r.next_u64();
  return ZKP_OK;
}
int32_t zkp_fs_rng_rand_u128(zkp_fs_rng* g, uint64_t out[2]) {
  if (!g || !out) return ZKP_ERR_BAD_ARG;
  out[0] = g->r.next_u64();                                  // rand 0.7: low half first
  out[1] = g->r.next_u64();
  return ZKP_OK;
}
int32_t zkp_fs_rng_rand_fr(zkp_fs_rng* g, zkp_curve_t curve, uint64_t* out_mont) {
  if (!g || !out_mont || (curve != ZKP_BN254 && curve != ZKP_BLS12_381)) return ZKP_ERR_BAD_ARG;
  g->rand_fr(curve, reinterpret_cast<uint32_t*>(out_mont));
  return ZKP_OK;
}
int32_t zkp_fs_rng_sample_outside_domain(zkp_fs_rng* g, zkp_curve_t curve, uint32_t log_domain, uint64_t* out_mont) {
  if (!g || !out_mont || (curve != ZKP_BN254 && curve != ZKP_BLS12_381) || log_domain > 32) return ZKP_ERR_BAD_ARG;
  const FrDesc f = fr_desc(curve);
  uint32_t t[8], p[8];
  for (;;) {
    g->rand_fr(curve, t);
    memcpy(p, t, 32);
    for (uint32_t k = 0; k < log_domain; k++) mont_mul(f, p, p, p);     // t^(2^k): vanishing polynomial + 1
    if (memcmp(p, f.one, 32) != 0) break;
  }
  memcpy(out_mont, t, 32);
  return ZKP_OK;
}
int32_t zkp_merlin_oneshot(const uint8_t* label, size_t label_len, const uint8_t* msg_label, size_t msg_label_len,
                           const uint8_t* msg, size_t msg_len, const uint8_t* chal_label, size_t chal_label_len,
                           uint8_t* out, size_t out_len) {
  if (!label || !msg_label || (msg_len && !msg) || !chal_label || !out) return ZKP_ERR_BAD_ARG;
  MerlinTranscript t(label, label_len);
  t.append_message(msg_label, msg_label_len, msg, msg_len);
  t.challenge_bytes(chal_label, chal_label_len, out, out_len);
  return ZKP_OK;
}

}  // extern "C"
