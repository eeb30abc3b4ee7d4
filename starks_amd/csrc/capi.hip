// capi.hip -- host side of libstarkhip.so: context, twiddle-table plans, the NTT / LDE / Merkle / FRI
// drivers and the C ABI declared in include/starkhip.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <utility>
#include <string>
#include <vector>

#include "../../include/starkhip.h"
#include "internal.hpp"
#include "knobs.hpp"

namespace {

constexpr int DIRECT_TABLE_LOG = 18;  // power tables up to 2^18 entries (8 MiB, L2-resident) are stored in full; larger ones as two halves

// ---- host field helpers (the same fp256.cuh code the device runs) -----------------------------------
fp h_from_wire(const uint8_t b[32]) {
  uint32_t w[8];
  memcpy(w, b, 32);
  return fp_canon(fp_from_wire_words(w));
}
void h_to_wire(const fp& a, uint8_t b[32]) {
  uint32_t w[8];
  fp_to_wire_words(fp_canon(a), w);
  memcpy(b, w, 32);
}
fp h_pow(fp a, uint64_t e) { return fp_pow_u64(a, e); }
fp h_inv(const fp& a) {  // a^(p-2); modp.py:71-79 uses extended Euclid, the residue is the same
  static const uint32_t e[8] = {0xffffffffu, 0xfffffea0u, 0xffffffffu, 0xffffffffu,
                                0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // p - 2
  fp r = fp_one(), b = a;
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 32] >> (i % 32)) & 1) r = fp_mul(r, b);
    b = fp_sqr(b);
  }
  return r;
}
fp h_pow_limbs(const fp& a, const uint32_t e[8]) {
  fp r = fp_one(), b = a;
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 32] >> (i % 32)) & 1) r = fp_mul(r, b);
    b = fp_sqr(b);
  }
  return r;
}
// 7^((p - 1) / 2^lg): the reference's choice of generator everywhere (stark.py:205, test_fft.py:120)
fp h_root_of_order_pow2(int lg) {
  const uint32_t pm1[8] = {0u, 0xfffffea1u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) {
    const int lo = i + lg / 32, sh = lg % 32;
    uint64_t v = lo < 8 ? pm1[lo] : 0;
    if (sh) v = (v >> sh) | ((uint64_t)(lo + 1 < 8 ? pm1[lo + 1] : 0) << (32 - sh));
    e[i] = (uint32_t)v;
  }
  return h_pow_limbs(fp_from_u32(7u), e);
}
int ilog2(uint64_t n) {
  int k = 0;
  while ((1ull << k) < n) ++k;
  return k;
}
bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }

// ---- plans ------------------------------------------------------------------------------------------
struct PowTable {  // g^e for e < order: lo[e & mask] * hi[e >> lb]; hi == nullptr when stored in full
  fp* lo = nullptr;
  fp* hi = nullptr;
  uint32_t lb = 0;
};

struct NttPlan {
  uint64_t n = 0;
  int log_n = 0;
  bool scaled = false;       // multiply by n^-1 (inverse transform)
  fp root;                   // effective root (already inverted for inverse transforms)
  std::vector<int> radix;    // log2 radix of each pass
  std::vector<const fp2*> wR;  // per pass: powers of root^(n/R), R/2 entries, as (w, w 2^128) pairs
  std::vector<fp*> tw2;      // per column pass: the same twiddles as rows, tw2[k * S + j2] = g^(j2 k) (null: table too large)
  std::vector<PowTable> tw;  // per column pass d: table of root^(P_d) (times n^-1 on pass 0 when scaled)
  PowTable base;             // unscaled table of root (sh_power_cycle, FRI fold)
  fp* scale = nullptr;       // n^-1 on the device (one-pass scaled plans)
  std::vector<void*> owned;  // device allocations to free
  size_t bytes = 0;          // sum of the owned allocations (plan-cache budget)
  uint64_t last_use = 0;     // ctx tick of the most recent lookup (LRU eviction)
};

}  // namespace

struct sh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t io_in = nullptr, io_out = nullptr;  // copy streams of the pipelined host-buffer transforms (created on first use)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t io_ev = nullptr;  // orders sh_dev_download_async's copies behind the ctx stream
  std::string err;
  std::map<std::string, NttPlan*> plans;
  // plan cache accounting: least-recently-used plans are evicted on entry of a public call once the tables held exceed
  // the byte budget (STARKHIP_PLAN_CACHE_MB, default 16 GiB of the 288 GB) or the plan count its cap
  size_t plan_bytes = 0, plan_budget = (size_t)16 << 30;
  uint64_t tick = 0, plans_built = 0, plans_evicted = 0;
  enum {
    WS_WIRE = 0, WS_X, WS_Y, WS_NTT, WS_TREE_A, WS_TREE_B, WS_COL_A, WS_COL_B, WS_MISC, WS_PROOF,
    WS_ST_TRACE, WS_ST_P, WS_ST_D, WS_ST_B, WS_ST_Q, WS_ST_SMALL, WS_ST_MTREE, WS_COUNT
  };
  void* ws[WS_COUNT] = {};
  size_t ws_cap[WS_COUNT] = {};
  // pinned staging for host <-> device copies of caller (pageable) buffers: two slots, double buffered
  uint8_t* pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  // STARK prover state: 1/((x_i - 1)(x_i - x_last)) and 1/(omega^j - 1) per (steps, ext); the step-polynomial terms last
  // uploaded (and their partial derivatives); the constraint flag
  std::map<std::pair<uint64_t, uint32_t>, void*> inv_z2;  // the three domain tables of (steps, ext): [3][n] (shk_stark_domain_tables)
  std::map<std::pair<uint64_t, uint32_t>, void*> inv_omega;
  std::vector<uint8_t> terms_key;
  void* terms_dev = nullptr;   // [terms][derivative terms]: see TermLayout
  uint32_t terms_begin[SHK_STARK_MAX_WIDTH + 1] = {};
  uint32_t terms_degree = 0;
  uint32_t* bad_flag = nullptr;  // [bad_cap] per-proof constraint flags of the calls since the last sh_stark_status*
  uint32_t bad_cap = 0;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      char buf_[256];                                                                        \
      snprintf(buf_, sizeof buf_, "%s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); \
      (ctx)->err = buf_;                                                                     \
      return e_ == hipErrorOutOfMemory ? SH_ERR_NOMEM : SH_ERR_HIP;                          \
    }                                                                                        \
  } while (0)

#define SH_TRY(expr)              \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != SH_OK) return rc_; \
  } while (0)

int ws_get(sh_ctx* c, int slot, size_t bytes, void** out) {
  if (bytes == 0) bytes = 32;
  if (bytes > c->ws_cap[slot]) {
    if (c->ws[slot]) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      HIP_TRY(c, hipFree(c->ws[slot]));
      c->ws[slot] = nullptr;
      c->ws_cap[slot] = 0;
    }
    size_t cap = (bytes + 4095) & ~(size_t)4095;
    HIP_TRY(c, hipMalloc(&c->ws[slot], cap));
    c->ws_cap[slot] = cap;
  }
  *out = c->ws[slot];
  return SH_OK;
}


constexpr size_t PIN_CHUNK = (size_t)8 << 20;

int pin_init(sh_ctx* c) {
  if (c->pin[0]) return SH_OK;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin[i]), PIN_CHUNK, hipHostMallocDefault));
    HIP_TRY(c, hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
  }
  return SH_OK;
}
int pin_wait(sh_ctx* c, int slot) {
  if (c->pin_busy[slot]) {
    HIP_TRY(c, hipEventSynchronize(c->pin_ev[slot]));
    c->pin_busy[slot] = false;
  }
  return SH_OK;
}
// caller memory the DMA engines can reach directly (sh_host_alloc, hipHostMalloc / hipHostRegister): no staging copy
bool host_is_pinned(const void* h) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, h) != hipSuccess) {
    (void)hipGetLastError();  // an unregistered pointer is an "error" to this query, not to us
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// host -> device on the ctx stream.  Pageable sources go through the pinned slots (the host memcpy of chunk i+1 overlaps
// the DMA of chunk i) and have been consumed when this returns.  A PINNED source of 64 KiB or more is handed to the DMA
// engine as it is and is read in stream order, i.e. possibly AFTER this returns: the caller must not reuse or free it
// before the stream has passed the copy (every entry point that takes caller buffers synchronises before it returns).
// `staged`: always go through the pinned slots, whatever the size (sources that die right after the call: plan tables).
int h2d(sh_ctx* c, void* d, const void* h, size_t bytes, bool staged = false) {
  if (bytes == 0) return SH_OK;
  if (!staged && bytes >= ((size_t)64 << 10) && host_is_pinned(h)) {
    HIP_TRY(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    return SH_OK;
  }
  if (!staged && bytes < ((size_t)64 << 10)) {
    HIP_TRY(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    return SH_OK;
  }
  SH_TRY(pin_init(c));
  size_t off = 0;
  for (int i = 0; off < bytes; ++i) {
    const int slot = i & 1;
    const size_t len = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
    SH_TRY(pin_wait(c, slot));
    memcpy(c->pin[slot], static_cast<const uint8_t*>(h) + off, len);
    HIP_TRY(c, hipMemcpyAsync(static_cast<uint8_t*>(d) + off, c->pin[slot], len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->pin_ev[slot], c->stream));
    c->pin_busy[slot] = true;
    off += len;
  }
  return SH_OK;
}
// device -> host (pageable); blocks until `h` is complete.
int d2h(sh_ctx* c, void* h, const void* d, size_t bytes) {
  if (bytes == 0) return SH_OK;
  if (bytes < ((size_t)64 << 10) || host_is_pinned(h)) {
    HIP_TRY(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SH_OK;
  }
  SH_TRY(pin_init(c));
  SH_TRY(pin_wait(c, 0));
  SH_TRY(pin_wait(c, 1));
  size_t off = 0, prev_off = 0, prev_len = 0;
  int prev_slot = -1;
  for (int i = 0; off < bytes; ++i) {
    const int slot = i & 1;
    const size_t len = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
    HIP_TRY(c, hipMemcpyAsync(c->pin[slot], static_cast<const uint8_t*>(d) + off, len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->pin_ev[slot], c->stream));
    if (prev_slot >= 0) {  // drain the previous chunk while this one is in flight
      HIP_TRY(c, hipEventSynchronize(c->pin_ev[prev_slot]));
      memcpy(static_cast<uint8_t*>(h) + prev_off, c->pin[prev_slot], prev_len);
    }
    prev_slot = slot;
    prev_off = off;
    prev_len = len;
    off += len;
  }
  HIP_TRY(c, hipEventSynchronize(c->pin_ev[prev_slot]));
  memcpy(static_cast<uint8_t*>(h) + prev_off, c->pin[prev_slot], prev_len);
  return SH_OK;
}

// device allocation owned by (and accounted to) a plan
int plan_alloc(sh_ctx* c, NttPlan* pl, size_t bytes, void** out) {
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, bytes ? bytes : 32));
  pl->owned.push_back(d);
  pl->bytes += bytes;
  *out = d;
  return SH_OK;
}
// host table -> device, in stream order (no device-wide synchronisation: the host vector is pageable, so h2d has
// consumed it on return, and every reader of the table is a later launch on the same stream)
int upload_bytes(sh_ctx* c, NttPlan* pl, const void* host, size_t bytes, void** dev) {
  SH_TRY(plan_alloc(c, pl, bytes, dev));
  return h2d(c, *dev, host, bytes, true);
}
int upload_table(sh_ctx* c, NttPlan* pl, const std::vector<fp>& host, fp** dev) {
  void* d = nullptr;
  SH_TRY(upload_bytes(c, pl, host.data(), host.size() * sizeof(fp), &d));
  *dev = reinterpret_cast<fp*>(d);
  return SH_OK;
}

// table of factor * g^e, e < 2^log_order (factor may be null).  Up to 2^direct_log entries the table is stored in full
// (one load per lookup); larger ones as two halves lo[e & mask] * hi[e >> lb] (one more modmul per lookup).  Full tables
// above 2^18 entries are expanded on the device from the two halves.
int build_pow_table(sh_ctx* c, NttPlan* pl, const fp& g, int log_order, const fp* factor, PowTable* out,
                    int direct_log = DIRECT_TABLE_LOG) {
  const bool direct = log_order <= direct_log;
  const bool host_direct = direct && log_order <= DIRECT_TABLE_LOG;
  const int lb = host_direct ? log_order : (log_order + 1) / 2;
  std::vector<fp> lo((size_t)1 << lb);
  lo[0] = fp_one();
  for (size_t i = 1; i < lo.size(); ++i) lo[i] = fp_mul(lo[i - 1], g);
  const fp gs = fp_mul(lo.back(), g);  // g^(2^lb)
  if (host_direct && factor)
    for (auto& v : lo) v = fp_mul(v, *factor);
  if (host_direct) {
    SH_TRY(upload_table(c, pl, lo, &out->lo));
    out->lb = (uint32_t)lb;
    out->hi = nullptr;
    return SH_OK;
  }
  std::vector<fp> hi((size_t)1 << (log_order - lb));
  hi[0] = factor ? *factor : fp_one();
  for (size_t i = 1; i < hi.size(); ++i) hi[i] = fp_mul(hi[i - 1], gs);
  if (!direct) {
    SH_TRY(upload_table(c, pl, lo, &out->lo));
    SH_TRY(upload_table(c, pl, hi, &out->hi));
    out->lb = (uint32_t)lb;
    return SH_OK;
  }
  // expand on the device, in stream order: full[e] = lo[e & mask] * hi[e >> lb] (the two small halves stay with the plan)
  void* full = nullptr;
  fp *dlo = nullptr, *dhi = nullptr;
  SH_TRY(plan_alloc(c, pl, sizeof(fp) << log_order, &full));
  SH_TRY(upload_table(c, pl, lo, &dlo));
  SH_TRY(upload_table(c, pl, hi, &dhi));
  HIP_TRY(c, shk_powers(dlo, dhi, (uint32_t)lb, reinterpret_cast<fp*>(full), (uint64_t)1 << log_order, c->stream));
  out->lo = reinterpret_cast<fp*>(full);
  out->hi = nullptr;
  out->lb = (uint32_t)log_order;
  return SH_OK;
}

// the passes of a 2^log_n-point transform (knobs.hpp:shk_choose_radices; one decomposition per size)
void choose_radices(int log_n, std::vector<int>* out) {
  int r[4];
  const int m = shk_choose_radices(log_n, r);
  out->assign(r, r + m);
}

constexpr size_t MAX_PLANS = 1024;  // count cap of the plan cache (the byte budget normally binds first)

void free_plan(sh_ctx* c, NttPlan* pl) {
  for (void* p : pl->owned) (void)hipFree(p);
  c->plan_bytes -= pl->bytes <= c->plan_bytes ? pl->bytes : c->plan_bytes;
  delete pl;
}

// Free every cached table (NTT plans, the STARK prover's inverse tables) and the workspaces, after the stream drained.
// Everything is rebuilt on demand.
int trim(sh_ctx* c) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto& kv : c->plans) free_plan(c, kv.second);
  c->plans.clear();
  c->plan_bytes = 0;
  for (auto& kv : c->inv_z2) (void)hipFree(kv.second);
  c->inv_z2.clear();
  for (auto& kv : c->inv_omega) (void)hipFree(kv.second);
  c->inv_omega.clear();
  for (int i = 0; i < sh_ctx::WS_COUNT; ++i) {
    if (c->ws[i]) (void)hipFree(c->ws[i]);
    c->ws[i] = nullptr;
    c->ws_cap[i] = 0;
  }
  return SH_OK;
}

// Least-recently-used plans go until the cache is inside its byte budget and count cap again; plans looked up recently
// (a prover's hot shapes) stay.  One stream synchronisation if anything is evicted (queued launches may read the tables).
int evict_plans(sh_ctx* c) {
  bool synced = false;
  while (!c->plans.empty() && (c->plan_bytes > c->plan_budget || c->plans.size() + 4 > MAX_PLANS)) {
    auto victim = c->plans.begin();
    for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
      if (it->second->last_use < victim->second->last_use) victim = it;
    if (!synced) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      synced = true;
    }
    free_plan(c, victim->second);
    c->plans.erase(victim);
    ++c->plans_evicted;
  }
  return SH_OK;
}

// Called on entry of every public function that builds plans, never while a plan pointer is held: a long-lived prover
// that meets many shapes must not grow for ever (a call creates at most 4 plans, so the budget can be overshot by one
// call's tables until the next entry).
int enter(sh_ctx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  return evict_plans(c);
}

std::string plan_key(const fp& root, uint64_t n, bool scaled) {
  uint8_t b[32];
  h_to_wire(root, b);
  std::string k(reinterpret_cast<const char*>(b), 32);
  k += std::to_string(n);
  k += scaled ? "s" : "u";
  return k;
}

// root must have order exactly n (a power of two): for n > 1 that is root^(n/2) == -1
// (the reference finds n by walking the powers of the root, fft.py:319-321).
int check_root_order(const fp& root, uint64_t n) {
  if (n == 1) return fp_eq_canon(fp_canon(root), fp_one()) ? SH_OK : SH_ERR_ROOT_ORDER;
  const fp h = fp_canon(h_pow(root, n / 2));
  const fp m1 = fp_canon(fp_neg(fp_one()));
  return fp_eq_canon(h, m1) ? SH_OK : SH_ERR_ROOT_ORDER;
}

int get_plan(sh_ctx* c, const fp& root_eff, uint64_t n, bool scaled, NttPlan** out) {
  std::vector<int> radix;
  choose_radices(ilog2(n), &radix);
  if (radix.empty()) return SH_ERR_UNSUPPORTED;
  const std::string key = plan_key(root_eff, n, scaled);
  auto it = c->plans.find(key);
  if (it != c->plans.end()) {
    it->second->last_use = ++c->tick;
    *out = it->second;
    return SH_OK;
  }
  // every exit before the plan is registered (error codes AND the early returns of HIP_TRY / SH_TRY) frees what was built
  struct Guard {
    sh_ctx* c;
    NttPlan* p;
    ~Guard() {
      if (p) {
        for (void* d : p->owned) (void)hipFree(d);
        delete p;
      }
    }
  } guard{c, nullptr};
  NttPlan* pl = new NttPlan();
  guard.p = pl;
  pl->n = n;
  pl->log_n = ilog2(n);
  pl->scaled = scaled;
  pl->root = root_eff;
  pl->radix = radix;
  const size_t m = pl->radix.size();
  fp ninv = fp_one();
  if (scaled) ninv = h_pow(h_inv(fp_from_u32(2u)), (uint64_t)pl->log_n);  // n^-1 = (2^-1)^log_n
  int rc = build_pow_table(c, pl, root_eff, pl->log_n, nullptr, &pl->base);
  if (rc == SH_OK && pl->log_n >= 2) {
    std::map<int, const fp2*> wr_by_radix;
    int log_P = 0;
    for (size_t d = 0; d < m && rc == SH_OK; ++d) {
      const int r = pl->radix[d];
      if (!wr_by_radix.count(r)) {
        const fp wr = h_pow(root_eff, n >> r);
        std::vector<fp> t((size_t)1 << (r - 1));
        t[0] = fp_one();
        for (size_t i = 1; i < t.size(); ++i) t[i] = fp_mul(t[i - 1], wr);
        // the butterflies multiply by these through fp_mul2: every entry is followed by its image times 2^128
        fp two128 = fp_zero();
        two128.v[4] = 1;
        std::vector<fp> pairs(2 * t.size());
        for (size_t i = 0; i < t.size(); ++i) {
          pairs[2 * i] = t[i];
          pairs[2 * i + 1] = fp_canon(fp_mul(t[i], two128));
        }
        fp* dev = nullptr;
        rc = upload_table(c, pl, pairs, &dev);
        wr_by_radix[r] = reinterpret_cast<const fp2*>(dev);
      }
      pl->wR.push_back(wr_by_radix[r]);
      if (rc == SH_OK && d + 1 < m) {
        PowTable t;
        if (d == 0 && !scaled) {
          t = pl->base;  // P_1 = 1: the table of the root itself
        } else {
          const fp g = h_pow(root_eff, 1ull << log_P);
          rc = build_pow_table(c, pl, g, pl->log_n - log_P, (d == 0 && scaled) ? &ninv : nullptr, &t);
        }
        pl->tw.push_back(t);
        // the tile passes read their inter-pass twiddles g^(j2 k) as rows of adjacent columns: a [k][j2] copy of the
        // table, tw2[k * S + j2] -- one coalesced load per element instead of a scattered one (plus, above 2^18
        // entries, the second modmul of the two-half lookup).  n / P entries: 32 MiB for the first pass of 2^20 points;
        // above 2^23 entries (STARKHIP_TW2_MAX_LOG) the pass keeps the power-table lookup.
        fp* tw2 = nullptr;
        const int log_S = pl->log_n - log_P - r;
        if (rc == SH_OK && r + log_S <= shk_knobs().tw2_max_log) {
          // a failed allocation of the row table (up to 512 MiB) is not fatal: the pass keeps the power-table lookup
          void* dv = nullptr;
          if (plan_alloc(c, pl, sizeof(fp) << (r + log_S), &dv) == SH_OK) {
            tw2 = reinterpret_cast<fp*>(dv);
            HIP_TRY(c, shk_tw2(t.lo, t.hi, t.lb, tw2, (uint32_t)r, (uint32_t)log_S, c->stream));
          } else {
            (void)hipGetLastError();
            c->err.clear();
          }
        }
        pl->tw2.push_back(tw2);
      }
      log_P += r;
    }
  }
  if (rc == SH_OK && scaled && m == 1) {
    std::vector<fp> s(1, ninv);
    rc = upload_table(c, pl, s, &pl->scale);
  }
  if (rc != SH_OK) return rc;  // the guard frees the partial plan
  guard.p = nullptr;
  pl->last_use = ++c->tick;
  c->plan_bytes += pl->bytes;
  ++c->plans_built;
  c->plans[key] = pl;
  *out = pl;
  return SH_OK;
}

// d_out = NTT(d_in) over plan->root; [batch][n] limb form; d_in may equal d_out.
// n_in != 0: d_in holds n_in <= n elements per vector ([batch][n_in]) and stands for its zero-padded extension
// (fft.py:323-324); d_in must then not alias d_out.  The first pass reads the short source and takes the rest as zeros.
int run_ntt(sh_ctx* c, NttPlan* pl, const fp* d_in, fp* d_out, uint32_t batch, uint64_t n_in = 0) {
  const uint64_t n = pl->n;
  if (batch == 0) return SH_OK;
  if (n_in >= n) n_in = 0;
  if (n_in && pl->log_n <= 1) {  // the tiny transform has no short-source load
    HIP_TRY(c, shk_pad_copy(d_in, d_out, n_in, n, batch, c->stream));
    d_in = d_out;
    n_in = 0;
  }
  if (pl->log_n <= 1) {
    HIP_TRY(c, shk_launch_ntt_tiny(d_in, d_out, (uint32_t)n, batch, n == 2 ? pl->scale : nullptr, c->stream));
    return SH_OK;
  }
  const size_t m = pl->radix.size();
  const fp* src = d_in;
  fp* work = d_out;
  if (m > 1) {
    void* w = nullptr;
    SH_TRY(ws_get(c, sh_ctx::WS_NTT, (size_t)batch * n * sizeof(fp), &w));
    work = reinterpret_cast<fp*>(w);
  }
  int log_P = 0;
  for (size_t d = 0; d < m; ++d) {
    const int r = pl->radix[d];
    NttPassArgs a;
    memset(&a, 0, sizeof a);
    a.log_n = (uint32_t)pl->log_n;
    a.wR = pl->wR[d];
    a.src_n = d == 0 ? n_in : 0;
    const bool last = d + 1 == m;
    if (!last) {
      a.src = src;
      a.dst = work;
      a.log_S = (uint32_t)(pl->log_n - log_P - r);
      a.total = (uint64_t)batch << (pl->log_n - r);  // batch * P * S columns
      a.tw_lo = pl->tw[d].lo;
      a.tw_hi = pl->tw[d].hi;
      a.tw_lb = pl->tw[d].lb;
      a.tw_direct = pl->tw[d].hi == nullptr;
      a.tw2 = pl->tw2[d];
      src = work;
    } else {
      a.src = src;
      a.dst = d_out;
      a.log_P = (uint32_t)(pl->log_n - r);
      a.total = (uint64_t)batch << a.log_P;  // rows
      a.ndig = (uint32_t)(m - 1);
      for (size_t k = 0; k + 1 < m; ++k) a.dig_log[k] = (uint32_t)pl->radix[k];
      a.scale = (m == 1) ? pl->scale : nullptr;
    }
    a.pass_index = (uint32_t)d;
    HIP_TRY(c, shk_launch_ntt_pass(r, last, a, c->stream));
    log_P += r;
  }
  return SH_OK;
}

// (`batch` = the vectors the caller is about to transform; no plan depends on it any more)
int plan_for(sh_ctx* c, const uint8_t root[32], uint64_t n, bool inverse, NttPlan** out, uint64_t batch = 0) {
  (void)batch;
  if (!is_pow2(n) || n > (1ull << 32)) return n > (1ull << 32) ? SH_ERR_UNSUPPORTED : SH_ERR_INVALID;
  fp w = h_from_wire(root);
  SH_TRY(check_root_order(w, n));
  if (inverse) w = h_pow(w, n - 1);  // w^-1: the reversed root list rootz[:0:-1] of fft.py:327
  return get_plan(c, w, n, inverse, out);
}

uint64_t fri_proof_len(uint64_t n, uint64_t maxdeg_plus_1, uint32_t samples) {
  uint64_t total = 0;
  bool first = true;
  while (maxdeg_plus_1 > 16 && n >= 16) {
    const uint64_t lg = (uint64_t)ilog2(n);
    total += 32 + (uint64_t)(first ? samples : 40) * 32 * ((lg - 1) + 4 * (lg + 1));
    n >>= 2;
    maxdeg_plus_1 >>= 2;
    first = false;
  }
  return total + 32 * n;
}

// every round's parameters are checked before anything is launched
int fri_validate(uint64_t n, uint64_t maxdeg_plus_1, uint32_t exclude, uint32_t samples) {
  uint64_t nn = n, md = maxdeg_plus_1;
  bool first = true;
  uint32_t rounds = 0;
  while (md > 16) {
    if (++rounds > SHK_FRI_MAX_ROUNDS) return SH_ERR_UNSUPPORTED;  // FriSampleArgs holds that many rounds (checked BEFORE any launch)
    if (nn < 16) return SH_ERR_INVALID;            // the reference cannot merkelize a column of < 4 values
    if ((nn >> 2) >= (1ull << 24)) return SH_ERR_UNSUPPORTED;  // assert modulus < 2**24 (utils.py:69)
    const uint32_t s = first ? samples : 40;
    if (s == 0) return SH_ERR_INVALID;
    if (exclude == 1) return SH_ERR_INVALID;       // division by zero in the reference (utils.py:90)
    if (exclude && ((nn >> 2) * (exclude - 1)) / exclude == 0) return SH_ERR_INVALID;
    nn >>= 2;
    md >>= 2;
    first = false;
  }
  return SH_OK;
}

// vals / tree: round 0 (the evaluations and their tree); next / tree2: the arenas of the later rounds, round r >= 1 at element
// offset batch * (n/4 + n/16 + ... + n/4^(r-1)) -- every round's column and tree stay until the commit's single sampling +
// gather pass at the end.
struct FriBuffers {
  fp *vals, *next;
  uint32_t *tree, *tree2, *ys;
};
int fri_buffers(sh_ctx* c, uint64_t n, uint32_t batch, uint32_t samples, FriBuffers* b) {
  void *va, *vb, *ta, *tb, *misc;
  SH_TRY(ws_get(c, sh_ctx::WS_COL_A, (size_t)batch * n * sizeof(fp), &va));
  SH_TRY(ws_get(c, sh_ctx::WS_COL_B, (size_t)batch * (n / 3 + 2) * sizeof(fp), &vb));
  SH_TRY(ws_get(c, sh_ctx::WS_TREE_A, (size_t)batch * 2 * n * 32, &ta));
  SH_TRY(ws_get(c, sh_ctx::WS_TREE_B, (size_t)batch * 2 * (n / 3 + 2) * 32, &tb));
  SH_TRY(ws_get(c, sh_ctx::WS_MISC, (size_t)batch * ((samples > 40 ? samples : 40) + 40 * SHK_FRI_MAX_ROUNDS) * 4 + 64, &misc));
  b->vals = reinterpret_cast<fp*>(va);
  b->next = reinterpret_cast<fp*>(vb);
  b->tree = reinterpret_cast<uint32_t*>(ta);
  b->tree2 = reinterpret_cast<uint32_t*>(tb);
  b->ys = reinterpret_cast<uint32_t*>(misc);
  return SH_OK;
}

// The rounds of the FRI commit (fri.py:212-266) on the evaluations in fb.vals (and, when have_tree, their Merkle
// tree in fb.tree); proof b is written at d_proof + b * stride.
int fri_rounds(sh_ctx* c, NttPlan* pl, FriBuffers fb, uint64_t n, uint64_t maxdeg_plus_1, uint32_t exclude, uint32_t samples,
               uint32_t batch, uint8_t* d_proof, uint64_t stride, bool have_tree) {
  fp* vals = fb.vals;
  fp* next = fb.next;
  uint32_t* tree = fb.tree;
  uint32_t* tree2 = fb.tree2;
  uint64_t nn = n, md = maxdeg_plus_1, off = 0;
  uint32_t round = 0;
  const fp inv_i = h_pow(pl->root, 3 * (n / 4));  // I^-1 = I^3, I = root^(n/4)
  FriSampleArgs sa;
  memset(&sa, 0, sizeof sa);
  sa.batch = batch;
  sa.exclude = exclude;
  sa.ys = fb.ys;
  sa.proof = d_proof;
  sa.proof_stride = stride;
  uint32_t ys_off = 0;
  while (md > 16) {  // at most SHK_FRI_MAX_ROUNDS rounds: fri_validate has refused anything longer before the first launch
    const uint32_t s = round == 0 ? samples : 40;
    if (!have_tree) HIP_TRY(c, shk_merkelize(vals, false, nn, batch, tree, c->stream, false));  // m = merkelize(values), fri.py:224
    FoldArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.values = vals;
    fa.nodes = tree;
    fa.column = next;
    fa.n = nn;
    fa.batch = batch;
    fa.tw_lo = pl->base.lo;
    fa.tw_hi = pl->base.hi;
    fa.tw_lb = pl->base.lb;
    fa.log_n0 = (uint32_t)pl->log_n;
    fa.round_shift = 2 * round;
    fa.inv_i = inv_i;
    HIP_TRY(c, shk_fri_fold_and_tree(fa, tree2, c->stream));                    // column and m2, fri.py:235-243
    // fri.py:246-254 (the 40 sampled rows and their 5 branches each): recorded, done for all rounds at the end
    const uint64_t lg = (uint64_t)ilog2(nn);
    FriRound& r = sa.r[round];
    r.values = vals;
    r.column = next;
    r.nodes_m = tree;
    r.nodes_m2 = tree2;
    r.n = nn;
    r.round_off = off;
    r.samples = s;
    r.ys_off = ys_off;
    r.work_begin = sa.work_total;
    sa.work_total += ((uint64_t)s * ((lg - 1) + 4 * (lg + 1)) + 1) * batch;
    ys_off += batch * s;
    off += 32 + (uint64_t)s * 32 * ((lg - 1) + 4 * (lg + 1));
    // Next round (fri.py:260-266): the reference inverse-transforms the column over root^4 and transforms it
    // back, which is the identity on the column; its tree m of round r+1 is this round's m2.  The column and its tree
    // stay where they are (the arenas), the round after writes behind them.
    vals = next;
    tree = tree2;
    next = next + (size_t)batch * (nn / 4);
    tree2 = tree2 + (size_t)batch * 2 * (nn / 4) * 8;
    have_tree = true;
    nn >>= 2;
    md >>= 2;
    ++round;
  }
  sa.rounds = round;
  sa.final_values = vals;  // fri.py:212-214
  sa.final_n = nn;
  sa.final_off = off;
  HIP_TRY(c, shk_fri_sample_and_gather_all(sa, c->stream));
  return SH_OK;
}

// FRI commit on device-resident coefficients; see starkhip.h for the layout.
int run_fri(sh_ctx* c, const fp* d_coeffs, uint64_t n, const uint8_t root[32], uint64_t maxdeg_plus_1,
            uint32_t exclude, uint32_t samples, uint32_t batch, uint8_t* d_proof, uint64_t n_coeffs = 0) {
  if (!d_coeffs || !d_proof || batch == 0 || !is_pow2(n)) return SH_ERR_INVALID;
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, false, &pl, batch));
  SH_TRY(fri_validate(n, maxdeg_plus_1, exclude, samples));
  FriBuffers fb;
  SH_TRY(fri_buffers(c, n, batch, samples, &fb));
  // values = fft(f) over the whole domain (fri.py:207-208)
  SH_TRY(run_ntt(c, pl, d_coeffs, fb.vals, batch, n_coeffs));  // n_coeffs != 0: [batch][n_coeffs], zero padding implicit
  return fri_rounds(c, pl, fb, n, maxdeg_plus_1, exclude, samples, batch, d_proof,
                    fri_proof_len(n, maxdeg_plus_1, samples), false);
}

// ---- STARK.mk_proof (stark.py:233-279) -----------------------------------------------------------------
uint64_t stark_header_len(uint64_t n, uint32_t width, uint32_t samples) {
  const uint64_t lg = (uint64_t)ilog2(n), k = 3ull * width;
  return 64 + (uint64_t)samples * 32 * (2 * (2 * k + (lg - 1)) + (lg + 1));
}

int stark_check_shape(uint64_t steps, uint32_t ext, uint32_t width, uint32_t degree, uint32_t samples) {
  if (!is_pow2(steps) || !is_pow2(ext) || steps < 2 || ext < 2 || width == 0 || samples == 0) return SH_ERR_INVALID;
  if (width > SHK_STARK_MAX_WIDTH) return SH_ERR_UNSUPPORTED;
  if (steps > (1ull << 24) || steps * ext >= (1ull << 24)) return SH_ERR_UNSUPPORTED;  // utils.py:69 (spot-check sampling)
  if ((uint64_t)degree * (steps - 1) + 1 >= steps * ext) return SH_ERR_UNSUPPORTED;   // C (X - x_last) must fit the domain
  return fri_validate(steps * ext, steps * degree, ext, 40);
}

// Device layout of the step-polynomial description (one allocation, ctx->terms_dev)
struct TermLayout {
  static constexpr size_t MAXT = SHK_STARK_MAX_TERMS, MAXD = SHK_STARK_MAX_TERMS * SHK_STARK_MAX_WIDTH;
  static constexpr size_t coef = 0;                                  // fp[MAXT]
  static constexpr size_t dcoef = coef + MAXT * sizeof(fp);          // fp[MAXD]
  static constexpr size_t ROW = SHK_STARK_MAX_WIDTH + 1;             // exponent rows: width bytes + 1 flag (coef == 1)
  static constexpr size_t exps = dcoef + MAXD * sizeof(fp);          // u8[MAXT][width + 1]
  static constexpr size_t dexps = exps + MAXT * ROW;                 // u8[MAXD][width + 1]
  static constexpr size_t dbegin = (dexps + MAXD * ROW + 3) & ~(size_t)3;  // u32[W * W + 1]
  static constexpr size_t total = dbegin + 4 * (SHK_STARK_MAX_WIDTH * SHK_STARK_MAX_WIDTH + 1);
};

// Parse + upload the step polynomials' terms and their partial derivatives (skipped when equal to the previous call's).
int stark_terms(sh_ctx* c, uint32_t width, const uint8_t* coefs, const uint8_t* exps, const uint32_t* counts) {
  if (!coefs || !exps || !counts) return SH_ERR_INVALID;
  uint64_t total = 0;
  for (uint32_t d = 0; d < width; ++d) total += counts[d];
  if (total == 0 || total > SHK_STARK_MAX_TERMS) return total ? SH_ERR_UNSUPPORTED : SH_ERR_INVALID;
  std::vector<uint8_t> key;
  key.push_back((uint8_t)width);
  key.insert(key.end(), reinterpret_cast<const uint8_t*>(counts), reinterpret_cast<const uint8_t*>(counts + width));
  key.insert(key.end(), coefs, coefs + 32 * total);
  key.insert(key.end(), exps, exps + (size_t)width * total);
  if (c->terms_dev && key == c->terms_key) return SH_OK;
  uint32_t degree = 0;
  for (uint64_t t = 0; t < total; ++t) {
    uint32_t sum = 0;
    for (uint32_t v = 0; v < width; ++v) sum += exps[t * width + v];
    if (sum > degree) degree = sum;  // MultivariatePolynomial.degree (multivariate_polynomial.py:111-117)
  }
  std::vector<uint8_t> img(TermLayout::total, 0);
  fp* cf = reinterpret_cast<fp*>(img.data() + TermLayout::coef);
  fp* dcf = reinterpret_cast<fp*>(img.data() + TermLayout::dcoef);
  uint8_t* ex = img.data() + TermLayout::exps;
  uint8_t* dex = img.data() + TermLayout::dexps;
  uint32_t* dbeg = reinterpret_cast<uint32_t*>(img.data() + TermLayout::dbegin);
  const fp one = fp_one();
  const size_t row = width + 1;
  for (uint64_t t = 0; t < total; ++t) {
    cf[t] = h_from_wire(coefs + 32 * t);
    memcpy(ex + t * row, exps + t * width, width);
    ex[t * row + width] = fp_eq_canon(cf[t], one) ? 1 : 0;
  }
  // d/dX_v of coef * prod X^e = (coef * e_v) * X_v^(e_v - 1) * prod_{u != v} X_u^e_u
  uint32_t begin[SHK_STARK_MAX_WIDTH + 1] = {0};
  for (uint32_t d = 0; d < width; ++d) begin[d + 1] = begin[d] + counts[d];
  uint32_t nd = 0;
  for (uint32_t d = 0; d < width; ++d) {
    for (uint32_t v = 0; v < width; ++v) {
      dbeg[d * width + v] = nd;
      for (uint32_t t = begin[d]; t < begin[d + 1]; ++t) {
        const uint32_t e = exps[(size_t)t * width + v];
        if (!e) continue;
        dcf[nd] = fp_canon(fp_mul(cf[t], fp_from_u32(e)));
        memcpy(dex + (size_t)nd * row, exps + (size_t)t * width, width);
        dex[(size_t)nd * row + v] = (uint8_t)(e - 1);
        dex[(size_t)nd * row + width] = fp_eq_canon(dcf[nd], one) ? 1 : 0;
        ++nd;
      }
    }
  }
  dbeg[width * width] = nd;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // earlier launches may still read the old terms
  if (!c->terms_dev) HIP_TRY(c, hipMalloc(&c->terms_dev, TermLayout::total));
  HIP_TRY(c, hipMemcpy(c->terms_dev, img.data(), TermLayout::total, hipMemcpyHostToDevice));
  memcpy(c->terms_begin, begin, sizeof begin);
  c->terms_degree = degree;
  c->terms_key.swap(key);
  return SH_OK;
}

// per-proof constraint flags: grown (and zeroed) on demand; flags already raised survive a growth
int bad_flags(sh_ctx* c, uint32_t batch) {
  if (batch <= c->bad_cap) return SH_OK;
  const uint32_t cap = (batch + 63u) & ~63u;
  uint32_t* nf = nullptr;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&nf), (size_t)cap * 4));
  hipError_t e = hipMemsetAsync(nf, 0, (size_t)cap * 4, c->stream);
  if (e == hipSuccess && c->bad_flag)
    e = hipMemcpyAsync(nf, c->bad_flag, (size_t)c->bad_cap * 4, hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    (void)hipFree(nf);
    HIP_TRY(c, e);
  }
  if (c->bad_flag) (void)hipFree(c->bad_flag);
  c->bad_flag = nf;
  c->bad_cap = cap;
  return SH_OK;
}

int run_stark(sh_ctx* c, fp* d_wit, const fp* d_inputs, uint64_t steps, uint32_t ext, uint32_t width, uint32_t samples,
              uint32_t batch, uint8_t* d_proof) {
  const uint32_t degree = c->terms_degree;
  SH_TRY(stark_check_shape(steps, ext, width, degree, samples));
  if (!d_wit || !d_inputs || !d_proof || batch == 0) return SH_ERR_INVALID;
  if (batch > 65535) return SH_ERR_UNSUPPORTED;  // the leaf / spot-check kernels launch one grid row (blockIdx.y) per proof
  const uint64_t n = steps * ext, cols = (uint64_t)batch * width;
  if (cols > 0xffffffffull) return SH_ERR_UNSUPPORTED;
  const fp g2 = h_root_of_order_pow2(ilog2(n));           // stark.py:205
  uint8_t g2b[32], g1b[32];
  h_to_wire(g2, g2b);
  h_to_wire(h_pow(g2, ext), g1b);                          // G1 = G2^ext (stark.py:208)
  NttPlan *fwd_n, *inv_s, *fwd_s;
  SH_TRY(plan_for(c, g2b, n, false, &fwd_n, cols));
  SH_TRY(plan_for(c, g1b, steps, true, &inv_s, cols));
  SH_TRY(plan_for(c, g1b, steps, false, &fwd_s, cols));
  const fp g1 = h_pow(g2, ext);
  const fp x_last = h_pow(g2, (steps - 1) * ext);          // stark.py:212
  const fp inv_last_m1 = h_inv(fp_sub(x_last, fp_one()));
  const fp cpow = h_pow(h_pow(g2, steps), n - 1);          // `powers[i]` after the loop: (G2^steps)^(precision-1) (stark.py:150-158)

  void *pe, *dw, *bw, *qv, *small, *mt;
  SH_TRY(ws_get(c, sh_ctx::WS_ST_P, cols * n * sizeof(fp), &pe));
  SH_TRY(ws_get(c, sh_ctx::WS_ST_D, cols * n * sizeof(fp), &dw));
  SH_TRY(ws_get(c, sh_ctx::WS_ST_B, cols * n * sizeof(fp), &bw));
  SH_TRY(ws_get(c, sh_ctx::WS_ST_Q, cols * steps * sizeof(fp), &qv));
  SH_TRY(ws_get(c, sh_ctx::WS_ST_MTREE, (size_t)batch * 2 * n * 32, &mt));
  const size_t iab_bytes = cols * 3 * sizeof(fp), scal_bytes = cols * 3 * sizeof(fp2);  // scalars as (s, s 2^128) pairs
  SH_TRY(ws_get(c, sh_ctx::WS_ST_SMALL, iab_bytes + scal_bytes + (size_t)batch * samples * 4, &small));
  fp* iab = reinterpret_cast<fp*>(small);
  fp* scal = reinterpret_cast<fp*>(reinterpret_cast<uint8_t*>(small) + iab_bytes);
  uint32_t* ys = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(small) + iab_bytes + scal_bytes);
  SH_TRY(bad_flags(c, batch));
  // cached per (steps, ext): 1 / (omega^j - 1) for the ext-th roots of unity omega^j = x^steps, and the domain tables
  // 1 / ((x_i - 1)(x_i - x_last)), x_i, (x_i - x_last) / (x_i^steps - 1)   (3 n elements, shared by every proof)
  fp* inv_omega = nullptr;
  {
    const auto key = std::make_pair(steps, ext);
    auto it = c->inv_omega.find(key);
    if (it == c->inv_omega.end()) {
      std::vector<fp> host(ext, fp_zero());
      const fp omega = h_pow(g2, steps);
      fp w = omega;
      for (uint32_t j = 1; j < ext; ++j) {
        host[j] = h_inv(fp_sub(w, fp_one()));
        w = fp_mul(w, omega);
      }
      void* t = nullptr;
      HIP_TRY(c, hipMalloc(&t, ext * sizeof(fp)));
      const hipError_t e = hipMemcpy(t, host.data(), ext * sizeof(fp), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(t);
        HIP_TRY(c, e);
      }
      c->inv_omega[key] = t;
      inv_omega = reinterpret_cast<fp*>(t);
    } else {
      inv_omega = reinterpret_cast<fp*>(it->second);
    }
  }
  fp* inv_z2 = nullptr;
  {
    const auto key = std::make_pair(steps, ext);
    auto it = c->inv_z2.find(key);
    if (it == c->inv_z2.end()) {
      void* t = nullptr;
      HIP_TRY(c, hipMalloc(&t, 3 * n * sizeof(fp)));
      inv_z2 = reinterpret_cast<fp*>(t);
      const hipError_t e = shk_stark_domain_tables(inv_z2, n, ext, fwd_n->base.lo, fwd_n->base.hi, fwd_n->base.lb, x_last, inv_omega,
                                                   c->stream);
      if (e != hipSuccess) {  // never cache a table whose fill did not launch
        (void)hipFree(t);
        HIP_TRY(c, e);
      }
      c->inv_z2[key] = t;
    } else {
      inv_z2 = reinterpret_cast<fp*>(it->second);
    }
  }
  const uint8_t* tb = reinterpret_cast<const uint8_t*>(c->terms_dev);
  StarkArgs a;
  memset(&a, 0, sizeof a);
  a.p_evals = reinterpret_cast<fp*>(pe);
  a.d_work = reinterpret_cast<fp*>(dw);
  a.b_work = reinterpret_cast<fp*>(bw);
  a.q_evals = reinterpret_cast<fp*>(qv);
  a.wit = d_wit;
  a.iab = iab;
  a.n = n;
  a.steps = steps;
  a.ext = ext;
  a.width = width;
  a.batch = batch;
  a.tw_lo = fwd_n->base.lo;
  a.tw_hi = fwd_n->base.hi;
  a.tw_lb = fwd_n->base.lb;
  a.inv_z2 = inv_z2;
  a.xpow = inv_z2 + n;
  a.fz = inv_z2 + 2 * n;
  a.inv_omega = inv_omega;
  a.x_last = x_last;
  a.g1 = g1;
  a.inv_steps = h_pow(h_inv(fp_from_u32(2u)), (uint64_t)ilog2(steps));
  a.inv_1_m_last = fp_neg(inv_last_m1);
  a.bad = c->bad_flag;
  a.term_coef = reinterpret_cast<const fp*>(tb + TermLayout::coef);
  a.term_exps = tb + TermLayout::exps;
  memcpy(a.term_begin, c->terms_begin, sizeof a.term_begin);
  a.dterm_coef = reinterpret_cast<const fp*>(tb + TermLayout::dcoef);
  a.dterm_exps = tb + TermLayout::dexps;
  a.dterm_begin = reinterpret_cast<const uint32_t*>(tb + TermLayout::dbegin);
  fp* P = reinterpret_cast<fp*>(pe);
  fp* Q = reinterpret_cast<fp*>(qv);

  // boundary interpolants need witness[dim][-1] before the trace becomes coefficients (stark.py:91-96)
  HIP_TRY(c, shk_stark_interp(d_wit, d_inputs, steps, (uint32_t)cols, inv_last_m1, iab, c->stream));
  // trace polynomials and their evaluations: the low-degree extension (stark.py:27-36, 253-256)
  // (coefficients go to the Q buffer: the witness stays what it is -- the trace polynomials' values on the trace points,
  // which the trace-point kernel reads contiguously)
  SH_TRY(run_ntt(c, inv_s, d_wit, Q, (uint32_t)cols));
  SH_TRY(run_ntt(c, fwd_n, Q, P, (uint32_t)cols, steps));  // the zero padding of fft_1d is implicit (fft.py:323-324)
  // Q = X P'(X) on the trace points, for the quotients' values there
  HIP_TRY(c, shk_stark_qprep(Q, Q, steps, cols, c->stream));
  SH_TRY(run_ntt(c, fwd_s, Q, Q, (uint32_t)cols));
  // D = C / Z and B = (P - I) / Z2 (stark.py:38-104), evaluated on the whole domain, and
  // mtree = merkelize_polynomial_evaluations(width, P + D + B evaluations) (stark.py:257)
  uint32_t* mtree = reinterpret_cast<uint32_t*>(mt);
  HIP_TRY(c, shk_stark_quotients_and_merkelize(a, mtree, c->stream));
  // l = pseudorandom linear combination keyed by mtree's root (stark.py:128-177, 259-263), on evaluations
  FriBuffers fb;
  SH_TRY(fri_buffers(c, n, batch, samples, &fb));
  HIP_TRY(c, shk_stark_scalars(mtree, 2 * n * 8, width, batch, cpow, scal, c->stream));
  HIP_TRY(c, shk_stark_lincomb_tree(a, scal, fb.vals, fb.tree, c->stream));  // l and l_mtree = merkelize(l) in one pass
  // spot checks (stark.py:390-402)
  const uint64_t stride = stark_header_len(n, width, samples) + fri_proof_len(n, steps * (uint64_t)degree, 40);
  HIP_TRY(c, shk_sample_indices(fb.tree, 2 * n * 8, (uint32_t)n, batch, samples, ext, ys, c->stream));
  HIP_TRY(c, shk_stark_gather(a, mtree, fb.tree, fb.vals, ys, samples, d_proof, stride, c->stream));
  // fri.generate_proximity_proof(l_poly, G2, steps * degree, exclude_multiples_of=ext) (stark.py:271-276); its first
  // tree is l_mtree
  return fri_rounds(c, fwd_n, fb, n, steps * (uint64_t)degree, ext, 40, batch, d_proof + stark_header_len(n, width, samples),
                    stride, true);
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* sh_strerror(int status) {
  switch (status) {
    case SH_OK: return "ok";
    case SH_ERR_INVALID: return "invalid argument";
    case SH_ERR_ROOT_ORDER: return "root_of_unity does not have order n";
    case SH_ERR_HIP: return "HIP runtime error";
    case SH_ERR_NOMEM: return "out of memory";
    case SH_ERR_TOO_SMALL: return "output buffer too small";
    case SH_ERR_UNSUPPORTED: return "unsupported size";
    case SH_ERR_CONSTRAINT: return "the witness violates a transition constraint";
    case SH_ERR_NO_DEVICE: return "no usable gfx950 device";
    case SH_ERR_REJECTED: return "proof rejected";
    default: return "unknown status";
  }
}
const char* sh_version(void) { return "starkhip 0.1 (gfx950)"; }

int sh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sh_ctx_create(int device, sh_ctx** out) {
  if (!out) return SH_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SH_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SH_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SH_ERR_NO_DEVICE;  // the code objects are gfx950 only
  sh_ctx* c = new sh_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SH_ERR_HIP;
  }
  if (shk_knobs().plan_cache_mb >= 0) c->plan_budget = (size_t)shk_knobs().plan_cache_mb << 20;
  *out = c;
  return SH_OK;
}

void sh_ctx_destroy(sh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)trim(c);
  if (c->terms_dev) (void)hipFree(c->terms_dev);
  if (c->bad_flag) (void)hipFree(c->bad_flag);
  for (int i = 0; i < 2; ++i) {
    if (c->pin[i]) (void)hipHostFree(c->pin[i]);
    if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
  }
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  if (c->io_in) (void)hipStreamSynchronize(c->io_in);
  if (c->io_out) (void)hipStreamSynchronize(c->io_out);
  if (c->io_ev) (void)hipEventDestroy(c->io_ev);
  if (c->io_in) (void)hipStreamDestroy(c->io_in);
  if (c->io_out) (void)hipStreamDestroy(c->io_out);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* sh_last_error(const sh_ctx* c) { return c ? c->err.c_str() : ""; }

int sh_sync(sh_ctx* c) {
  if (!c) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}
int sh_timer_start(sh_ctx* c) {
  if (!c) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  return SH_OK;
}
int sh_timer_stop(sh_ctx* c, float* ms) {
  if (!c || !ms) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  HIP_TRY(c, hipEventSynchronize(c->ev1));
  HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return SH_OK;
}

// ---- device-resident API ------------------------------------------------------------------------------
int sh_dev_alloc(sh_ctx* c, uint64_t bytes, void** dptr) {
  if (!c || !dptr) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMalloc(dptr, bytes ? bytes : 32));
  return SH_OK;
}
int sh_dev_free(sh_ctx* c, void* dptr) {
  if (!c) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipFree(dptr));
  return SH_OK;
}
int sh_dev_upload(sh_ctx* c, const void* host_src, void* d_dst, uint64_t bytes) {
  if (!c || (!host_src && bytes) || (!d_dst && bytes)) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  SH_TRY(h2d(c, d_dst, host_src, bytes));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}
int sh_dev_download(sh_ctx* c, const void* d_src, void* host_dst, uint64_t bytes) {
  if (!c || (!host_dst && bytes) || (!d_src && bytes)) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  return d2h(c, host_dst, d_src, bytes);
}
int sh_host_alloc(sh_ctx* c, uint64_t bytes, void** hptr) {
  if (!c || !hptr) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipHostMalloc(hptr, bytes ? bytes : 32, hipHostMallocDefault));
  return SH_OK;
}
int sh_host_free(sh_ctx* c, void* hptr) {
  if (!c) {
    // the context that allocated it is gone (sh_ctx_destroy frees no caller buffers): a pinned buffer belongs to the process,
    // so it can still be released -- device-wide, since no stream is left to drain
    if (!hptr) return SH_OK;
    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    return hipHostFree(hptr) == hipSuccess ? SH_OK : SH_ERR_HIP;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipHostFree(hptr));
  return SH_OK;
}
int sh_dev_download_async(sh_ctx* c, const void* d_src, void* host_dst, uint64_t bytes) {
  if (!c || (bytes && (!d_src || !host_dst))) return SH_ERR_INVALID;
  if (!bytes) return SH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!host_is_pinned(host_dst)) return SH_ERR_INVALID;  // a pageable destination would make the copy synchronous and staged
  if (!c->io_out) HIP_TRY(c, hipStreamCreateWithFlags(&c->io_out, hipStreamNonBlocking));
  if (!c->io_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->io_ev, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(c->io_ev, c->stream));           // behind the work queued so far ...
  HIP_TRY(c, hipStreamWaitEvent(c->io_out, c->io_ev, 0));    // ... (the wait is enqueued, the event may be re-recorded at once)
  HIP_TRY(c, hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->io_out));
  return SH_OK;
}
int sh_io_sync(sh_ctx* c) {
  if (!c) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->io_in) HIP_TRY(c, hipStreamSynchronize(c->io_in));
  if (c->io_out) HIP_TRY(c, hipStreamSynchronize(c->io_out));
  return SH_OK;
}
int sh_dev_copy(sh_ctx* c, const void* d_src, void* d_dst, uint64_t bytes) {
  if (!c || (bytes && (!d_src || !d_dst))) return SH_ERR_INVALID;
  if (!bytes) return SH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return SH_OK;
}
int sh_dev_download_2d(sh_ctx* c, const void* d_src, uint64_t src_pitch, void* host_dst, uint64_t width, uint64_t rows) {
  if (!c || !d_src || !host_dst || width > src_pitch) return SH_ERR_INVALID;
  if (!width || !rows) return SH_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy2DAsync(host_dst, width, d_src, src_pitch, width, rows, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}
int sh_dev_from_wire(sh_ctx* c, const uint8_t* host_wire, void* d_limbs, uint64_t n) {
  if (!c || (n && (!host_wire || !d_limbs))) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  void* w = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, n * 32, &w));
  SH_TRY(h2d(c, w, host_wire, n * 32));
  HIP_TRY(c, shk_wire_to_limb(reinterpret_cast<const uint8_t*>(w), reinterpret_cast<fp*>(d_limbs), n, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}
int sh_dev_to_wire(sh_ctx* c, const void* d_limbs, uint8_t* host_wire, uint64_t n) {
  if (!c || (n && (!host_wire || !d_limbs))) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  void* w = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, n * 32, &w));
  HIP_TRY(c, shk_limb_to_wire(reinterpret_cast<const fp*>(d_limbs), reinterpret_cast<uint8_t*>(w), n, c->stream));
  return d2h(c, host_wire, w, n * 32);
}
int sh_dev_fill_seeded(sh_ctx* c, void* d_limbs, uint64_t n, uint64_t seed) {
  if (!c || (n && !d_limbs)) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, shk_fill_seeded(reinterpret_cast<fp*>(d_limbs), n, seed, c->stream));
  return SH_OK;
}
int sh_dev_ntt(sh_ctx* c, const void* d_in, void* d_out, uint64_t n, uint32_t batch, const uint8_t root[32],
               int inverse) {
  if (!c || !d_in || !d_out || !root) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, inverse != 0, &pl, batch));
  return run_ntt(c, pl, reinterpret_cast<const fp*>(d_in), reinterpret_cast<fp*>(d_out), batch);
}
int sh_dev_lde(sh_ctx* c, void* d_trace, void* d_out, uint64_t steps, uint32_t ext, uint32_t cols, const uint8_t g2[32]) {
  if (!c || !d_trace || !d_out || !g2 || cols == 0 || !is_pow2(steps) || !is_pow2(ext)) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  const uint64_t n = steps * ext;
  NttPlan *inv1 = nullptr, *fwd2 = nullptr;
  SH_TRY(plan_for(c, g2, n, false, &fwd2, cols));
  uint8_t g1b[32];
  h_to_wire(h_pow(fwd2->root, ext), g1b);  // G1 = G2^ext (stark.py:217-220)
  SH_TRY(plan_for(c, g1b, steps, true, &inv1, cols));
  fp* t = reinterpret_cast<fp*>(d_trace);
  fp* x = reinterpret_cast<fp*>(d_out);
  SH_TRY(run_ntt(c, inv1, t, t, cols));                                  // stark.py:27-36
  return run_ntt(c, fwd2, t, x, cols, steps);                            // stark.py:253-256; fft_1d's zero padding implicit
}
int sh_dev_merkelize(sh_ctx* c, const void* d_values, uint64_t n, uint32_t batch, void* d_nodes) {
  if (!c || !d_values || !d_nodes || !is_pow2(n) || n < 4 || batch == 0) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, shk_merkelize(d_values, false, n, batch, reinterpret_cast<uint32_t*>(d_nodes), c->stream));
  return SH_OK;
}
int sh_dev_fri_fold(sh_ctx* c, const void* d_values, const void* d_nodes, uint64_t n, uint32_t batch,
                    const uint8_t root[32], void* d_column) {
  if (!c || !d_values || !d_nodes || !d_column || !root || n < 4) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, false, &pl));
  FoldArgs fa;
  memset(&fa, 0, sizeof fa);
  fa.values = reinterpret_cast<const fp*>(d_values);
  fa.nodes = reinterpret_cast<const uint32_t*>(d_nodes);
  fa.column = reinterpret_cast<fp*>(d_column);
  fa.n = n;
  fa.batch = batch;
  fa.tw_lo = pl->base.lo;
  fa.tw_hi = pl->base.hi;
  fa.tw_lb = pl->base.lb;
  fa.log_n0 = (uint32_t)pl->log_n;
  fa.round_shift = 0;
  fa.inv_i = h_pow(pl->root, 3 * (n / 4));
  HIP_TRY(c, shk_fri_fold(fa, c->stream));
  return SH_OK;
}
int sh_dev_fri_prove(sh_ctx* c, const void* d_coeffs, uint64_t n, const uint8_t root[32], uint64_t maxdeg_plus_1,
                     uint32_t exclude, uint32_t samples, uint32_t batch, void* d_proof) {
  if (!c || !root) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  return run_fri(c, reinterpret_cast<const fp*>(d_coeffs), n, root, maxdeg_plus_1, exclude, samples, batch,
                 reinterpret_cast<uint8_t*>(d_proof));
}

int sh_dev_fri_prove_coeffs(sh_ctx* c, const void* d_coeffs, uint64_t n_coeffs, uint64_t n, const uint8_t root[32],
                            uint64_t maxdeg_plus_1, uint32_t exclude, uint32_t samples, uint32_t batch, void* d_proof) {
  if (!c || !root || n_coeffs == 0 || n_coeffs > n) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  return run_fri(c, reinterpret_cast<const fp*>(d_coeffs), n, root, maxdeg_plus_1, exclude, samples, batch,
                 reinterpret_cast<uint8_t*>(d_proof), n_coeffs);
}

// ---- host-buffer API --------------------------------------------------------------------------------
static int upload_padded(sh_ctx* c, const uint8_t* in, uint64_t n_in, uint64_t n, uint32_t batch, int slot, fp** out) {
  void *w = nullptr, *x = nullptr, *y = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, (size_t)batch * (n_in > n ? n_in : n) * 32, &w));
  SH_TRY(ws_get(c, slot, (size_t)batch * n * sizeof(fp), &x));
  if (n_in == n) {
    SH_TRY(h2d(c, w, in, (size_t)batch * n * 32));
    HIP_TRY(c, shk_wire_to_limb(reinterpret_cast<uint8_t*>(w), reinterpret_cast<fp*>(x), (uint64_t)batch * n, c->stream));
  } else {
    SH_TRY(ws_get(c, sh_ctx::WS_MISC, (size_t)batch * (n_in ? n_in : 1) * sizeof(fp), &y));
    if (n_in) {
      SH_TRY(h2d(c, w, in, (size_t)batch * n_in * 32));
      HIP_TRY(c, shk_wire_to_limb(reinterpret_cast<uint8_t*>(w), reinterpret_cast<fp*>(y), (uint64_t)batch * n_in, c->stream));
    }
    HIP_TRY(c, shk_pad_copy(reinterpret_cast<fp*>(y), reinterpret_cast<fp*>(x), n_in, n, batch, c->stream));  // fft.py:323-324
  }
  *out = reinterpret_cast<fp*>(x);
  return SH_OK;
}
// 0 < n_in < n: the values as they are ([batch][n_in] in WS_Y, a slot no FRI / NTT driver touches); the transform's first pass takes the zero padding as
// implicit (run_ntt's n_in).  Otherwise the padded array of upload_padded, *n_short = 0.
static int upload_short(sh_ctx* c, const uint8_t* in, uint64_t n_in, uint64_t n, uint32_t batch, int slot, fp** out,
                        uint64_t* n_short) {
  *n_short = 0;
  if (n_in == 0 || n_in >= n) return upload_padded(c, in, n_in, n, batch, slot, out);
  void *w = nullptr, *y = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, (size_t)batch * n * 32, &w));  // sized for the n-point result's download as well
  SH_TRY(ws_get(c, sh_ctx::WS_Y, (size_t)batch * n_in * sizeof(fp), &y));
  SH_TRY(h2d(c, w, in, (size_t)batch * n_in * 32));
  HIP_TRY(c, shk_wire_to_limb(reinterpret_cast<uint8_t*>(w), reinterpret_cast<fp*>(y), (uint64_t)batch * n_in, c->stream));
  *out = reinterpret_cast<fp*>(y);
  *n_short = n_in;
  return SH_OK;
}
static int download_wire(sh_ctx* c, const fp* d, uint8_t* out, uint64_t count) {
  void* w = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, (size_t)count * 32, &w));
  HIP_TRY(c, shk_limb_to_wire(d, reinterpret_cast<uint8_t*>(w), count, c->stream));
  return d2h(c, out, w, (size_t)count * 32);
}

// Several vectors from page-locked host buffers: the vectors go through in up to 8 chunks, H2D on one copy stream, wire -> limb,
// transform, limb -> wire on the ctx stream, D2H on a second copy stream, chained by events -- the upload of chunk k + 1 and the
// download of chunk k - 1 run under the transform of chunk k (PCIe is full duplex), instead of upload, transform, download in a row.
static int ntt_batch_pipelined(sh_ctx* c, NttPlan* pl, const uint8_t* in, uint8_t* out, uint64_t n, uint32_t batch) {
  const uint32_t nch = batch < 8 ? batch : 8;
  const uint32_t per = (batch + nch - 1) / nch;
  void *w_in = nullptr, *x = nullptr, *w_out = nullptr, *ntt_ws = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, (size_t)batch * n * 32, &w_in));
  SH_TRY(ws_get(c, sh_ctx::WS_X, (size_t)batch * n * sizeof(fp), &x));
  SH_TRY(ws_get(c, sh_ctx::WS_Y, (size_t)batch * n * 32, &w_out));
  SH_TRY(ws_get(c, sh_ctx::WS_NTT, (size_t)per * n * sizeof(fp), &ntt_ws));  // sized once: run_ntt must not reallocate mid-pipeline
  if (!c->io_in) HIP_TRY(c, hipStreamCreateWithFlags(&c->io_in, hipStreamNonBlocking));
  if (!c->io_out) HIP_TRY(c, hipStreamCreateWithFlags(&c->io_out, hipStreamNonBlocking));
  struct Events {
    hipEvent_t e[17] = {};
    ~Events() {
      for (hipEvent_t x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } ev;
  for (uint32_t k = 0; k < 2 * nch + 1; ++k) HIP_TRY(c, hipEventCreateWithFlags(&ev.e[k], hipEventDisableTiming));
  // earlier work on the ctx stream may still use the workspaces: the uploads start behind it
  HIP_TRY(c, hipEventRecord(ev.e[2 * nch], c->stream));
  HIP_TRY(c, hipStreamWaitEvent(c->io_in, ev.e[2 * nch], 0));
  int rc = SH_OK;
  for (uint32_t k = 0; k < nch && rc == SH_OK; ++k) {
    const uint64_t v0 = (uint64_t)k * per, v1 = v0 + per < batch ? v0 + per : batch;
    if (v0 >= v1) break;
    const size_t off = (size_t)v0 * n * 32, len = (size_t)(v1 - v0) * n * 32;
    hipError_t e = hipMemcpyAsync(static_cast<uint8_t*>(w_in) + off, in + off, len, hipMemcpyHostToDevice, c->io_in);
    if (e == hipSuccess) e = hipEventRecord(ev.e[k], c->io_in);
    if (e != hipSuccess) rc = SH_ERR_HIP;
  }
  for (uint32_t k = 0; k < nch && rc == SH_OK; ++k) {
    const uint64_t v0 = (uint64_t)k * per, v1 = v0 + per < batch ? v0 + per : batch;
    if (v0 >= v1) break;
    const size_t off = (size_t)v0 * n * 32, len = (size_t)(v1 - v0) * n * 32;
    fp* xk = reinterpret_cast<fp*>(x) + v0 * n;
    hipError_t e = hipStreamWaitEvent(c->stream, ev.e[k], 0);
    if (e == hipSuccess) e = shk_wire_to_limb(static_cast<uint8_t*>(w_in) + off, xk, (v1 - v0) * n, c->stream);
    if (e != hipSuccess) { rc = SH_ERR_HIP; break; }
    rc = run_ntt(c, pl, xk, xk, (uint32_t)(v1 - v0));
    if (rc != SH_OK) break;
    e = shk_limb_to_wire(xk, static_cast<uint8_t*>(w_out) + off, (v1 - v0) * n, c->stream);
    if (e == hipSuccess) e = hipEventRecord(ev.e[nch + k], c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->io_out, ev.e[nch + k], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(out + off, static_cast<uint8_t*>(w_out) + off, len, hipMemcpyDeviceToHost, c->io_out);
    if (e != hipSuccess) rc = SH_ERR_HIP;
  }
  // whatever happened, nothing of this call may still be in flight when the caller's buffers and the events go away
  (void)hipStreamSynchronize(c->io_in);
  (void)hipStreamSynchronize(c->stream);
  const hipError_t es = hipStreamSynchronize(c->io_out);
  if (rc == SH_OK && es != hipSuccess) rc = SH_ERR_HIP;
  if (rc == SH_ERR_HIP && c->err.empty()) c->err = "pipelined transform: HIP error";
  return rc;
}

int sh_ntt_batch(sh_ctx* c, const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t n, uint32_t batch,
                 const uint8_t root[32], int inverse) {
  if (!c || !out || !root || (n_in && !in) || batch == 0) return SH_ERR_INVALID;
  if (n_in > n) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, inverse != 0, &pl, batch));
  if (batch >= 2 && n_in == n && n >= (1u << 12) && (size_t)batch * n * 32 >= ((size_t)4 << 20) && host_is_pinned(in) &&
      host_is_pinned(out))
    return ntt_batch_pipelined(c, pl, in, out, n, batch);
  fp* x = nullptr;
  uint64_t n_short = 0;
  SH_TRY(upload_short(c, in, n_in, n, batch, sh_ctx::WS_X, &x, &n_short));
  if (n_short) {
    void* y = nullptr;
    SH_TRY(ws_get(c, sh_ctx::WS_X, (size_t)batch * n * sizeof(fp), &y));
    SH_TRY(run_ntt(c, pl, x, reinterpret_cast<fp*>(y), batch, n_short));  // fft.py:323-324, zeros not materialised
    x = reinterpret_cast<fp*>(y);
  } else {
    SH_TRY(run_ntt(c, pl, x, x, batch));
  }
  return download_wire(c, x, out, (uint64_t)batch * n);
}
int sh_ntt(sh_ctx* c, const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t n, const uint8_t root[32], int inverse) {
  return sh_ntt_batch(c, in, n_in, out, n, 1, root, inverse);
}

int sh_mul_polys(sh_ctx* c, const uint8_t* a, uint64_t n_a, const uint8_t* b, uint64_t n_b, uint8_t* out, uint64_t n,
                 const uint8_t root[32]) {
  if (!c || !out || !root || (n_a && !a) || (n_b && !b) || n_a > n || n_b > n) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan *fwd = nullptr, *rev = nullptr;
  SH_TRY(plan_for(c, root, n, false, &fwd, 1));
  SH_TRY(get_plan(c, h_pow(fwd->root, n - 1), n, false, &rev));  // reversed roots, NO 1/n (fft.py:345)
  fp *x = nullptr, *y = nullptr;
  SH_TRY(upload_padded(c, a, n_a, n, 1, sh_ctx::WS_X, &x));
  SH_TRY(upload_padded(c, b, n_b, n, 1, sh_ctx::WS_Y, &y));
  SH_TRY(run_ntt(c, fwd, x, x, 1));
  SH_TRY(run_ntt(c, fwd, y, y, 1));
  HIP_TRY(c, shk_pointwise_mul(x, y, x, n, c->stream));
  SH_TRY(run_ntt(c, rev, x, x, 1));
  return download_wire(c, x, out, n);
}

int sh_power_cycle(sh_ctx* c, const uint8_t root[32], uint64_t n, uint8_t* out) {
  if (!c || !out || !root) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, false, &pl));
  void* x = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_X, (size_t)n * sizeof(fp), &x));
  HIP_TRY(c, shk_powers(pl->base.lo, pl->base.hi, pl->base.lb, reinterpret_cast<fp*>(x), n, c->stream));
  return download_wire(c, reinterpret_cast<fp*>(x), out, n);
}

int sh_lde(sh_ctx* c, const uint8_t* trace, uint8_t* out, uint64_t steps, uint32_t ext, uint32_t cols,
           const uint8_t g2[32]) {
  if (!c || !trace || !out || !g2 || cols == 0 || !is_pow2(steps) || !is_pow2(ext)) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  const uint64_t n = steps * ext;
  NttPlan *inv1 = nullptr, *fwd2 = nullptr;
  SH_TRY(plan_for(c, g2, n, false, &fwd2, cols));
  uint8_t g1b[32];
  h_to_wire(h_pow(fwd2->root, ext), g1b);  // G1 = G2^ext (stark.py:217-220)
  SH_TRY(plan_for(c, g1b, steps, true, &inv1, cols));
  fp* t = nullptr;
  SH_TRY(upload_padded(c, trace, steps, steps, cols, sh_ctx::WS_Y, &t));
  SH_TRY(run_ntt(c, inv1, t, t, cols));  // trace polynomial coefficients (stark.py:27-36)
  void* x = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_X, (size_t)cols * n * sizeof(fp), &x));
  SH_TRY(run_ntt(c, fwd2, t, reinterpret_cast<fp*>(x), cols, steps));  // stark.py:253-256
  return download_wire(c, reinterpret_cast<fp*>(x), out, (uint64_t)cols * n);
}

int sh_merkelize(sh_ctx* c, const uint8_t* leaves, uint64_t n, uint8_t* nodes) {
  if (!c || !leaves || !nodes || !is_pow2(n) || n < 4) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  void *w = nullptr, *t = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, (size_t)n * 32, &w));
  SH_TRY(ws_get(c, sh_ctx::WS_TREE_A, (size_t)2 * n * 32, &t));
  SH_TRY(h2d(c, w, leaves, (size_t)n * 32));
  HIP_TRY(c, shk_merkelize(w, true, n, 1, reinterpret_cast<uint32_t*>(t), c->stream));
  return d2h(c, nodes, t, (size_t)2 * n * 32);
}

int sh_merkelize_packed(sh_ctx* c, const uint8_t* evals, uint64_t n, uint32_t k, uint8_t* nodes, uint8_t* leaves) {
  if (!c || !evals || !nodes || !leaves || !is_pow2(n) || n < 4 || k == 0) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  void *w = nullptr, *t = nullptr, *l = nullptr;
  const size_t ebytes = (size_t)n * k * 32;
  SH_TRY(ws_get(c, sh_ctx::WS_WIRE, ebytes, &w));
  SH_TRY(ws_get(c, sh_ctx::WS_TREE_A, (size_t)2 * n * 32, &t));
  SH_TRY(ws_get(c, sh_ctx::WS_X, ebytes, &l));
  SH_TRY(h2d(c, w, evals, ebytes));
  HIP_TRY(c, shk_merkelize_packed(reinterpret_cast<const uint8_t*>(w), n, k, reinterpret_cast<uint8_t*>(l),
                                  reinterpret_cast<uint32_t*>(t), c->stream));
  SH_TRY(d2h(c, nodes, t, (size_t)n * 32));
  return d2h(c, leaves, l, ebytes);
}

int sh_fri_fold(sh_ctx* c, const uint8_t* values, uint64_t n, const uint8_t root[32], const uint8_t special_x[32],
                uint8_t* column) {
  if (!c || !values || !root || !special_x || !column || !is_pow2(n) || n < 4) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  NttPlan* pl = nullptr;
  SH_TRY(plan_for(c, root, n, false, &pl));
  fp* v = nullptr;
  SH_TRY(upload_padded(c, values, n, n, 1, sh_ctx::WS_X, &v));
  void *col = nullptr, *sx = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_COL_B, (size_t)(n / 4) * sizeof(fp), &col));
  SH_TRY(ws_get(c, sh_ctx::WS_MISC, 64, &sx));
  HIP_TRY(c, hipMemcpyAsync(sx, special_x, 32, hipMemcpyHostToDevice, c->stream));
  FoldArgs fa;
  memset(&fa, 0, sizeof fa);
  fa.values = v;
  fa.nodes = nullptr;
  fa.special_x = reinterpret_cast<const uint32_t*>(sx);
  fa.column = reinterpret_cast<fp*>(col);
  fa.n = n;
  fa.batch = 1;
  fa.tw_lo = pl->base.lo;
  fa.tw_hi = pl->base.hi;
  fa.tw_lb = pl->base.lb;
  fa.log_n0 = (uint32_t)pl->log_n;
  fa.inv_i = h_pow(pl->root, 3 * (n / 4));
  HIP_TRY(c, shk_fri_fold(fa, c->stream));
  return download_wire(c, reinterpret_cast<fp*>(col), column, n / 4);
}

uint32_t sh_ntt_passes(uint64_t n, uint32_t batch) {
  if (!is_pow2(n) || n > (1ull << 32)) return 0;
  (void)batch;  // one decomposition per size
  int r[4];
  return (uint32_t)shk_choose_radices(ilog2(n), r);
}

uint64_t sh_fri_proof_len(uint64_t n, uint64_t maxdeg_plus_1, uint32_t samples) {
  return fri_proof_len(n, maxdeg_plus_1, samples);
}

int sh_fri_prove(sh_ctx* c, const uint8_t* coeffs, uint64_t n_coeffs, uint64_t n, const uint8_t root[32],
                 uint64_t maxdeg_plus_1, uint32_t exclude, uint32_t samples, uint32_t batch, uint8_t* proof,
                 uint64_t proof_cap) {
  if (!c || !root || !proof || (n_coeffs && !coeffs) || batch == 0 || n_coeffs > n || !is_pow2(n)) return SH_ERR_INVALID;
  SH_TRY(enter(c));
  const uint64_t stride = fri_proof_len(n, maxdeg_plus_1, samples);
  if (proof_cap < stride * batch) return SH_ERR_TOO_SMALL;
  fp* x = nullptr;
  uint64_t n_short = 0;
  SH_TRY(upload_short(c, coeffs, n_coeffs, n, batch, sh_ctx::WS_X, &x, &n_short));
  void* dp = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_PROOF, (size_t)stride * batch, &dp));
  SH_TRY(run_fri(c, x, n, root, maxdeg_plus_1, exclude, samples, batch, reinterpret_cast<uint8_t*>(dp), n_short));
  return d2h(c, proof, dp, (size_t)stride * batch);
}

uint64_t sh_stark_proof_len(uint64_t steps, uint32_t ext, uint32_t width, uint32_t degree, uint32_t samples) {
  if (stark_check_shape(steps, ext, width, degree, samples) != SH_OK) return 0;
  const uint64_t n = steps * ext;
  return stark_header_len(n, width, samples) + fri_proof_len(n, steps * (uint64_t)degree, 40);
}

int sh_stark_status_batch(sh_ctx* c, uint8_t* bad, uint32_t batch) {
  if (!c || (batch && !bad)) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (bad) memset(bad, 0, batch);
  if (!c->bad_flag) return SH_OK;
  std::vector<uint32_t> flags(c->bad_cap);
  HIP_TRY(c, hipMemcpy(flags.data(), c->bad_flag, (size_t)c->bad_cap * 4, hipMemcpyDeviceToHost));
  bool any = false;
  for (uint32_t b = 0; b < c->bad_cap; ++b) {
    if (!flags[b]) continue;
    any = true;
    if (b < batch) bad[b] = 1;
  }
  if (!any) return SH_OK;
  HIP_TRY(c, hipMemset(c->bad_flag, 0, (size_t)c->bad_cap * 4));
  return SH_ERR_CONSTRAINT;
}
int sh_stark_status(sh_ctx* c) { return sh_stark_status_batch(c, nullptr, 0); }

int sh_ctx_trim(sh_ctx* c) {
  if (!c) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  return trim(c);
}
int sh_ctx_set_plan_budget(sh_ctx* c, uint64_t bytes) {
  if (!c) return SH_ERR_INVALID;
  c->plan_budget = (size_t)bytes;
  return enter(c);
}
int sh_ctx_stats(const sh_ctx* c, uint64_t out[4]) {
  if (!c || !out) return SH_ERR_INVALID;
  out[0] = c->plans.size();
  out[1] = c->plan_bytes;
  out[2] = c->plans_built;
  out[3] = c->plans_evicted;
  return SH_OK;
}

int sh_dev_fill_mimc_units(sh_ctx* c, void* d_witness, void* d_inputs, uint64_t steps, uint32_t first_unit, uint32_t batch,
                           uint32_t constant) {
  if (!c || !d_witness || !d_inputs || steps == 0 || batch == 0) return SH_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, shk_fill_mimc_units(reinterpret_cast<fp*>(d_witness), reinterpret_cast<fp*>(d_inputs), steps, first_unit, batch,
                                 constant, c->stream));
  return SH_OK;
}

int sh_dev_stark_prove(sh_ctx* c, void* d_witness, const void* d_inputs, uint64_t steps, uint32_t ext, uint32_t width,
                       const uint8_t* term_coefs, const uint8_t* term_exps, const uint32_t* term_counts, uint32_t samples,
                       uint32_t batch, void* d_proof) {
  if (!c || width == 0 || width > SHK_STARK_MAX_WIDTH) return c && width > SHK_STARK_MAX_WIDTH ? SH_ERR_UNSUPPORTED : SH_ERR_INVALID;
  SH_TRY(enter(c));
  SH_TRY(stark_terms(c, width, term_coefs, term_exps, term_counts));
  return run_stark(c, reinterpret_cast<fp*>(d_witness), reinterpret_cast<const fp*>(d_inputs), steps, ext, width, samples,
                   batch, reinterpret_cast<uint8_t*>(d_proof));
}

int sh_stark_prove(sh_ctx* c, const uint8_t* witness, const uint8_t* inputs, uint64_t steps, uint32_t ext, uint32_t width,
                   const uint8_t* term_coefs, const uint8_t* term_exps, const uint32_t* term_counts, uint32_t samples,
                   uint32_t batch, uint8_t* proof, uint64_t proof_cap) {
  if (!c || !witness || !inputs || !proof || batch == 0 || width == 0) return SH_ERR_INVALID;
  if (width > SHK_STARK_MAX_WIDTH) return SH_ERR_UNSUPPORTED;
  SH_TRY(enter(c));
  SH_TRY(stark_terms(c, width, term_coefs, term_exps, term_counts));
  SH_TRY(stark_check_shape(steps, ext, width, c->terms_degree, samples));
  const uint64_t n = steps * ext;
  const uint64_t stride = stark_header_len(n, width, samples) + fri_proof_len(n, steps * (uint64_t)c->terms_degree, 40);
  if (proof_cap < stride * batch) return SH_ERR_TOO_SMALL;
  const uint64_t cols = (uint64_t)batch * width;
  fp *w = nullptr, *in = nullptr;
  SH_TRY(upload_padded(c, witness, steps, steps, (uint32_t)cols, sh_ctx::WS_ST_TRACE, &w));
  SH_TRY(upload_padded(c, inputs, 1, 1, (uint32_t)cols, sh_ctx::WS_Y, &in));
  void* dp = nullptr;
  SH_TRY(ws_get(c, sh_ctx::WS_PROOF, (size_t)stride * batch, &dp));
  // this call reports on its own witnesses only: flags left by unchecked sh_dev_stark_prove calls are dropped
  if (c->bad_flag) HIP_TRY(c, hipMemsetAsync(c->bad_flag, 0, (size_t)c->bad_cap * 4, c->stream));
  SH_TRY(run_stark(c, w, in, steps, ext, width, samples, batch, reinterpret_cast<uint8_t*>(dp)));
  SH_TRY(d2h(c, proof, dp, (size_t)stride * batch));
  return sh_stark_status(c);
}

}  // extern "C"
