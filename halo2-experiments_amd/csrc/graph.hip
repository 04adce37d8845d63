// graph.hip -- evaluate_h's gate arithmetic on the device (second half): the GraphEvaluator of
// halo2_proofs::plonk::evaluation (the crate pinned at /root/reference/Cargo.toml:10; reached from create_proof,
// /root/reference/src/circuits/utils.rs:40-48).  Upstream flattens every gate / lookup expression of a circuit
// into a straight-line program over "value sources" and runs it once per row of the extended domain on the CPU:
//
//     ValueSource   Constant(i) | Intermediate(i) | Fixed(col, rot) | Advice(col, rot) | Instance(col, rot)
//                   | Challenge(i) | Beta | Gamma | Theta | Y | PreviousValue
//     Calculation   Add | Sub | Mul | Square | Double | Negate | Horner(start, parts, factor) | Store
//     row index of a rotated query: (idx + rot * 2^(extended_k - k)) mod 2^extended_k
//
// Here: one lane per row, all columns resident in HBM (extended-domain arrays, external Montgomery words), the
// program interpreted instruction by instruction with wave-uniform decode.  Fixed / Advice / Instance are one
// column table; challenges, beta, gamma, theta, y are constants of the call (the mirror in evaluation.py maps
// them); Horner is lowered to MulAdd steps.  Intermediates live in a scratch array laid out [slot][word][lane] so
// that every access is coalesced; the host first maps the program's intermediates to slots by liveness (upstream
// gives every calculation a fresh index; only a few dozen are live at once), so the scratch stays L2-sized.
// Values are kept in ff29's internal form (< 3r, normalised limbs); a column word is converted on load.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>

#include "g1.h"
#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

constexpr int GE_THREADS = 256;

enum GraphOp : uint32_t { GOP_ADD = 0, GOP_SUB = 1, GOP_MUL = 2, GOP_SQUARE = 3, GOP_DOUBLE = 4, GOP_NEGATE = 5, GOP_STORE = 6, GOP_MULADD = 7 };
enum GraphSrc : uint32_t { GSRC_CONST = 0, GSRC_INTER = 1, GSRC_COLUMN = 2, GSRC_PREV = 3 };
// a source is one word: kind (bits 30..31) | rotation index (bits 20..29) | index (bits 0..19); a column source uses
// index bits 0..13 for the column and bits 14..19 for the column's log2 row count when it is SHORTER than the domain
// (read at row mod 2^that; 0 = a full-size column): the period lives in the instruction, not in a per-call table.
__host__ __device__ inline uint32_t gsrc_kind(uint32_t s) { return s >> 30; }
__host__ __device__ inline uint32_t gsrc_rot(uint32_t s) { return (s >> 20) & 1023u; }
__host__ __device__ inline uint32_t gsrc_index(uint32_t s) { return s & 0xfffffu; }
__host__ __device__ inline uint32_t gsrc_column(uint32_t s) { return s & 0x3fffu; }
__host__ __device__ inline uint32_t gsrc_log_rows(uint32_t s) { return (s >> 14) & 63u; }

struct GraphCalc {      // device form: 5 words; op carries the forwarding flags in bits 8..11
  uint32_t op, a, b, c, target;
};
// Expression trees are evaluated depth first, so the result of calculation k is very often an operand of k + 1 and of
// nothing else (45 % of all intermediate reads and 55 % of all writes of the MerkleSumTree program): the lowering marks
// those operands "take the previous result from registers" and those targets "never stored", which removes that share
// of the [slot][word][lane] scratch traffic -- the interpreter is bound by it, not by the arithmetic.
constexpr uint32_t GF_A_PREV = 1u << 8, GF_B_PREV = 1u << 9, GF_C_PREV = 1u << 10, GF_NO_STORE = 1u << 11;
// Lazy reductions.  An intermediate is a normalised element (29-bit limbs) of value < GE_CAP * r, not < 3r: products
// accept that (inputs < 18r give outputs < 3r; the column sums depend on the limbs only), so a sum or difference needs
// its ~50-instruction modular reduction only when its STATIC bound -- propagated through the program at lowering time --
// would pass GE_CAP.  GF_NO_REDUCE: carry propagation only.  GF_SUB_WIDE: the subtrahend may exceed 3r, so the
// subtraction adds 20r instead of 4r (and is always reduced).  Bounds of every combination: hc_graph_bounds_closure
// (host_check.cpp, HM_BOUNDS build), tests/test_ff29_host.py.
constexpr uint32_t GF_NO_REDUCE = 1u << 12, GF_SUB_WIDE = 1u << 13;
constexpr double GE_CAP = 16.0, GE_COLUMN_BOUND = 6.0;     // a packed 256-bit word is < 2^256 < 5.3 r whatever it holds

__device__ __forceinline__ Fr ge_reduce(const Fr& lazy) { return fe_reduce_small(fe_norm(lazy)); }   // any lazy sum < 2^261 -> < 3r

__device__ __forceinline__ Fr ge_from_ext(const uint32_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_mul(fe_unpack<FrParams>(w), fe_const<FrParams>(FrParams::EXT2INT));
}
// A column that already holds INTERNAL-form words (32 x the external value, canonical: what the coset NTT writes when
// its fused constants are pre-multiplied by 32) needs no conversion product: a third of the MerkleSumTree program's
// multiplications were conversions of column loads.
__device__ __forceinline__ Fr ge_from_internal(const uint32_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Fr r = fe_unpack<FrParams>(w);
  HM_DECLARE(r, GE_COLUMN_BOUND);
  return r;
}

// The call's column table and per-call constants (2.6 KiB) travel through a DEVICE buffer of the stream's AuxSlot,
// filled by a stream-ordered copy ahead of the launch.  Rounds 1-2 passed the struct by value, and a variant with a
// BYTE table in it (the columns' periods) aborted at run time.  What is known (tools/ubench/kernarg_byval.hip,
// profiles/r04_kernarg_byval.txt, profiles/r05_kernarg_isa.txt): not the argument's size (3 176 bytes of kernarg, limit
// 4 096; no scratch).  A dynamically indexed table of 32- or 64-bit entries in a by-value argument is read with SCALAR
// loads from the kernarg segment (s_load_dword s, s[0:1], s_off), which work; a table of BYTES cannot be (s_load has no
// sub-dword form), so the compiler emits a VECTOR load addressed through the kernarg pointer (global_load_ubyte v, v,
// s[0:1] offset:...).  A stand-alone kernel of that shape faulted ONCE on this stack and runs without the byte table --
// a correlation, not an established cause: the load is legal ISA, s[0:1] is intact and the offset in range.  HYPOTHESIS:
// the vector path cannot read the runtime's kernarg buffer here.  The library does not rest on it: this kernel takes ONE
// pointer to a device copy (every table access an ordinary global load, a 72-byte argument block whatever the program's
// shape), and tests/test_isa.py asserts on the built objects that no kernel of the library addresses a vector memory
// instruction through its kernarg pointer.  tests/test_evaluation.py runs 256 columns, short-period columns and 16
// per-call constants through it.
constexpr uint32_t GE_MAX_COLUMNS = 256;
constexpr uint32_t GE_MAX_DYN = 16;
struct GraphColumns {
  const uint32_t* p[GE_MAX_COLUMNS];
  uint32_t dyn[GE_MAX_DYN * 9];   // challenges, beta, gamma, theta, y ... of THIS proof, internal form
  uint32_t n_static;              // constants [0, n_static) come from the program, [n_static, ..) from dyn
};

template <bool INTERNAL>
__device__ __forceinline__ Fr ge_fetch(uint32_t src, const GraphColumns* __restrict__ columns, const uint32_t* __restrict__ consts,
                                       const int32_t* __restrict__ rotations, const uint32_t* __restrict__ scratch, uint32_t T,
                                       uint32_t lane_slot, uint64_t idx, uint64_t mask, const uint32_t* __restrict__ prev) {
  const uint32_t kind = gsrc_kind(src), index = gsrc_index(src);
  Fr r;
  if (kind == GSRC_INTER) {
    const uint32_t* p = scratch + (size_t)index * 9 * T + lane_slot;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[(size_t)i * T];
    HM_DECLARE(r, GE_CAP);
  } else if (kind == GSRC_CONST) {
    if (index >= columns->n_static) {
      const uint32_t d = (index - columns->n_static) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) r.l[i] = columns->dyn[d + i];
    } else {
      const uint32_t* p = consts + (size_t)index * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) r.l[i] = p[i];
    }
    HM_DECLARE(r, 1.0);
  } else if (kind == GSRC_COLUMN) {
    // two's complement: a negative rotation wraps -- inside the row's SEGMENT (mask = segment length - 1; one segment = the
    // whole domain in the ordinary call, one coset of the extended domain in hm_graph_evaluate_segments_dev)
    uint64_t row = (idx & ~mask) | ((idx + (uint64_t)(int64_t)rotations[gsrc_rot(src)]) & mask);
    const uint32_t lr = gsrc_log_rows(src);          // a short column (the vanishing polynomial's inverse pattern) is periodic
    if (lr != 0) row &= (1ull << lr) - 1ull;
    const uint32_t* cell = columns->p[gsrc_column(src)] + row * 8;
    r = INTERNAL ? ge_from_internal(cell) : ge_from_ext(cell);
  } else {
    r = ge_from_ext(prev);
  }
  return r;
}

template <bool INTERNAL>
__global__ __launch_bounds__(GE_THREADS) void graph_evaluate_kernel(const GraphColumns* __restrict__ columns,
                                                                    const uint32_t* __restrict__ consts,
                                                                    const int32_t* __restrict__ rotations,
                                                                    const GraphCalc* __restrict__ calcs, uint32_t n_calc, uint32_t result_src,
                                                                    uint32_t result_prev, uint32_t* __restrict__ scratch,
                                                                    uint32_t* __restrict__ values, uint64_t size, uint32_t log_segment) {
  const uint32_t T = gridDim.x * GE_THREADS;
  const uint32_t lane_slot = blockIdx.x * GE_THREADS + threadIdx.x;
  const uint64_t mask = (1ull << log_segment) - 1;
  for (uint64_t idx = lane_slot; idx < size; idx += T) {
    uint32_t* vrow = values + idx * 8;
    Fr prev = fe_zero<FrParams>();                         // the previous calculation's result, in registers
    HM_DECLARE(prev, 3.0);
    for (uint32_t k = 0; k < n_calc; ++k) {
      const GraphCalc cc = calcs[k];                       // the same words for every lane: scalar loads
      const uint32_t op = cc.op & 0xffu;
      auto src = [&](uint32_t word, uint32_t flag) -> Fr {
        if (cc.op & flag) return prev;                     // wave-uniform branch
        return ge_fetch<INTERNAL>(word, columns, consts, rotations, scratch, T, lane_slot, idx, mask, vrow);
      };
      const Fr a = src(cc.a, GF_A_PREV);
      Fr out;
      const bool lazy = (cc.op & GF_NO_REDUCE) != 0, wide = (cc.op & GF_SUB_WIDE) != 0;      // wave-uniform
      auto settle = [&](const Fr& t) -> Fr { return lazy ? fe_norm(t) : ge_reduce(t); };
      switch (op) {
        case GOP_ADD:
          out = settle(fe_add(a, src(cc.b, GF_B_PREV)));
          break;
        case GOP_SUB: {
          const Fr b = src(cc.b, GF_B_PREV);
          out = wide ? ge_reduce(fe_sub<20, 29>(a, b)) : settle(fe_sub<4, 29>(a, b));
          break;
        }
        case GOP_MUL:
          out = fe_mul(a, src(cc.b, GF_B_PREV));
          break;
        case GOP_SQUARE:
          out = fe_sqr(a);
          break;
        case GOP_DOUBLE:
          out = settle(fe_dbl(a));
          break;
        case GOP_NEGATE:
          out = wide ? ge_reduce(fe_sub<20, 29>(fe_zero<FrParams>(), a)) : settle(fe_sub<4, 29>(fe_zero<FrParams>(), a));
          break;
        case GOP_MULADD: {   // a * b + c (one Horner step)
          const Fr b = src(cc.b, GF_B_PREV);
          const Fr c = src(cc.c, GF_C_PREV);
          out = settle(fe_add(fe_mul(a, b), c));
          break;
        }
        default:             // GOP_STORE
          out = a;
          break;
      }
      if (!(cc.op & GF_NO_STORE)) {
        uint32_t* p = scratch + (size_t)cc.target * 9 * T + lane_slot;
#pragma unroll
        for (int i = 0; i < 9; ++i) p[(size_t)i * T] = out.l[i];
      }
      prev = out;
    }
    // the graph's value: its last calculation (upstream GraphEvaluator::evaluate), or the given source
    Fr res = fe_zero<FrParams>();
    if (result_prev)
      res = prev;
    else if (n_calc != 0 || gsrc_kind(result_src) != GSRC_INTER)
      res = ge_fetch<INTERNAL>(result_src, columns, consts, rotations, scratch, T, lane_slot, idx, mask, vrow);
    uint32_t w[8];
    fe_to_ext(w, ge_reduce(res));
    uint4* dst = reinterpret_cast<uint4*>(vrow);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// ---- host: program "compilation" (validation, Horner-free device form, liveness slot allocation) and launch ----
static bool src_ok(uint32_t s, size_t n_const, size_t n_inter, size_t n_cols, size_t n_rot) {
  switch (gsrc_kind(s)) {
    case GSRC_CONST: return gsrc_index(s) < n_const;
    case GSRC_INTER: return gsrc_index(s) < n_inter;
    case GSRC_COLUMN: return gsrc_column(s) < n_cols && gsrc_rot(s) < n_rot && gsrc_log_rows(s) <= 30;
    default: return true;
  }
}

// Lower the validated program for one column format.  Steps: (1) copy propagation -- a Store of a constant, or of a
// column when columns need no conversion, defines nothing new: its users read the source directly (upstream stores every
// queried cell once so that the CPU loop converts it once; here a column read costs what a scratch read costs);
// (2) forwarding flags and store elision (see GF_*); (3) linear-scan slot allocation by liveness for what is still stored.
static int graph_lower(const GraphProgram& g, bool internal_cols, GraphVariant& out) {
  const size_t n_in = g.calcs5.size() / 5;
  const uint32_t n_inter = g.n_intermediates;
  auto nsrc_of = [](uint32_t op) { return op == GOP_MULADD ? 3 : (op <= GOP_MUL ? 2 : 1); };
  // (1) copy propagation
  std::vector<uint32_t> alias(n_inter, 0xffffffffu);          // intermediate -> the source word that replaces it
  auto resolve = [&](uint32_t s) { return gsrc_kind(s) == GSRC_INTER && alias[gsrc_index(s)] != 0xffffffffu ? alias[gsrc_index(s)] : s; };
  struct Ins { uint32_t op, src[3], target; };
  std::vector<Ins> prog;
  prog.reserve(n_in);
  for (size_t k = 0; k < n_in; ++k) {
    const uint32_t* c = &g.calcs5[5 * k];
    Ins in{c[0], {resolve(c[1]), resolve(c[2]), resolve(c[3])}, c[4]};
    if (in.op == GOP_STORE) {
      const uint32_t kd = gsrc_kind(in.src[0]);
      if (kd == GSRC_CONST || (kd == GSRC_COLUMN && internal_cols)) {
        alias[in.target] = in.src[0];
        continue;
      }
    }
    prog.push_back(in);
  }
  uint32_t result_src = n_in ? resolve((GSRC_INTER << 30) | g.calcs5[5 * (n_in - 1) + 4]) : ((GSRC_INTER << 30) | 0u);
  const size_t n = prog.size();
  // (2) uses of every intermediate; forwarding and store elision
  std::vector<std::vector<uint32_t>> uses(n_inter);
  for (size_t k = 0; k < n; ++k)
    for (int j = 0; j < nsrc_of(prog[k].op); ++j)
      if (gsrc_kind(prog[k].src[j]) == GSRC_INTER) uses[gsrc_index(prog[k].src[j])].push_back((uint32_t)k);
  const bool result_is_inter = n_in != 0 && gsrc_kind(result_src) == GSRC_INTER;
  const bool result_prev = result_is_inter && n != 0 && prog[n - 1].target == gsrc_index(result_src);
  std::vector<uint32_t> flags(n, 0);
  std::vector<char> stored(n, 1);
  for (size_t k = 0; k < n; ++k) {
    if (k > 0)
      for (int j = 0; j < nsrc_of(prog[k].op); ++j)
        if (gsrc_kind(prog[k].src[j]) == GSRC_INTER && gsrc_index(prog[k].src[j]) == prog[k - 1].target)
          flags[k] |= j == 0 ? GF_A_PREV : (j == 1 ? GF_B_PREV : GF_C_PREV);
    bool only_next = true;
    for (uint32_t u : uses[prog[k].target]) only_next = only_next && u == (uint32_t)k + 1;
    const bool is_result = result_is_inter && prog[k].target == gsrc_index(result_src);
    if (only_next && (!is_result || (result_prev && k == n - 1))) {     // read (if at all) by the next instruction only
      stored[k] = 0;
      flags[k] |= GF_NO_STORE;
    }
  }
  // (2b) static value bounds (in units of r) -> which sums and differences keep their reduction
  {
    std::vector<double> vb(n_inter, 3.0);
    auto bound_of = [&](uint32_t sw) -> double {
      switch (gsrc_kind(sw)) {
        case GSRC_CONST: return 1.0;                                  // canonical constants
        case GSRC_INTER: return vb[gsrc_index(sw)];
        case GSRC_COLUMN: return internal_cols ? GE_COLUMN_BOUND : 3.0;   // external words pass through a product
        default: return 3.0;                                          // PreviousValue: a product output
      }
    };
    for (size_t k = 0; k < n; ++k) {
      const Ins& in = prog[k];
      double r = 3.0;
      uint32_t f = 0;
      switch (in.op) {
        case GOP_ADD: r = bound_of(in.src[0]) + bound_of(in.src[1]); break;
        case GOP_DOUBLE: r = 2.0 * bound_of(in.src[0]); break;
        case GOP_MULADD: r = 3.0 + bound_of(in.src[2]); break;
        case GOP_SUB:
        case GOP_NEGATE: {
          const double minuend = in.op == GOP_SUB ? bound_of(in.src[0]) : 0.0;
          const double subtrahend = bound_of(in.src[in.op == GOP_SUB ? 1 : 0]);
          if (subtrahend <= 3.0) {
            r = minuend + 4.0;
          } else {
            f |= GF_SUB_WIDE;                                         // reduced in the kernel whatever r is
            r = 1e9;
          }
          break;
        }
        case GOP_STORE: r = bound_of(in.src[0]); f |= GF_NO_REDUCE; break;     // a copy
        default: r = 3.0; f |= GF_NO_REDUCE; break;                   // Mul / Square: product outputs (the flag is not read)
      }
      if (!(f & (GF_NO_REDUCE | GF_SUB_WIDE))) {
        if (r <= GE_CAP) f |= GF_NO_REDUCE;
        else r = 3.0;                                                 // reduced
      } else if (f & GF_SUB_WIDE) {
        r = 3.0;
      }
      vb[in.target] = r;
      flags[k] |= f;
    }
  }
  // (3) slots for what is stored
  std::vector<uint32_t> last_use(n_inter, 0), slot_of(n_inter, 0xffffffffu), free_slots;
  for (size_t k = 0; k < n; ++k)
    for (int j = 0; j < nsrc_of(prog[k].op); ++j)
      if (gsrc_kind(prog[k].src[j]) == GSRC_INTER) last_use[gsrc_index(prog[k].src[j])] = (uint32_t)k;
  if (result_is_inter && !result_prev) last_use[gsrc_index(result_src)] = (uint32_t)n;
  std::multimap<uint32_t, uint32_t> expiring;                  // last use -> slot
  std::vector<GraphCalc> dev(n);
  uint32_t n_slots = 0;
  auto remap = [&](uint32_t s) -> uint32_t {
    if (gsrc_kind(s) != GSRC_INTER) return s;
    const uint32_t sl = slot_of[gsrc_index(s)];
    return (GSRC_INTER << 30) | (sl == 0xffffffffu ? 0u : sl);   // a never-stored operand is always taken from registers
  };
  for (size_t k = 0; k < n; ++k) {
    const Ins& in = prog[k];
    const int ns = nsrc_of(in.op);
    GraphCalc d{in.op | flags[k], remap(in.src[0]), ns > 1 ? remap(in.src[1]) : 0u, ns > 2 ? remap(in.src[2]) : 0u, 0};
    // slots whose value was read for the last time BEFORE this instruction are free (its own operands are read before
    // its target is written, so a slot expiring AT k may be reused as k's target)
    while (!expiring.empty() && expiring.begin()->first <= (uint32_t)k) {
      free_slots.push_back(expiring.begin()->second);
      expiring.erase(expiring.begin());
    }
    if (stored[k]) {
      uint32_t slot;
      if (!free_slots.empty()) {
        slot = free_slots.back();
        free_slots.pop_back();
      } else {
        slot = n_slots++;
      }
      slot_of[in.target] = slot;
      // a value that is never read again still needs its slot for this one instruction
      expiring.emplace(last_use[in.target] > (uint32_t)k ? last_use[in.target] : (uint32_t)k + 1, slot);
      d.target = slot;
    }
    dev[k] = d;
  }
  out.n_calc = (uint32_t)n;
  out.n_slots = n_slots ? n_slots : 1;
  out.result_prev = result_prev ? 1u : 0u;
  out.result_src = result_is_inter ? remap(result_src) : result_src;
  HM_HIP_CHECK(hipMalloc(&out.d_calcs, std::max<size_t>(n, 1) * sizeof(GraphCalc)));
  if (n) HM_HIP_CHECK(hipMemcpy(out.d_calcs, dev.data(), n * sizeof(GraphCalc), hipMemcpyHostToDevice));
  out.ready = true;
  return HM_OK;
}

int graph_create(DeviceCtx& ctx, const uint32_t* calcs5, size_t n_calc, const uint64_t* constants_ext, size_t n_const_static,
                 size_t n_dynamic, const int32_t* rotations, size_t n_rot, size_t n_columns, uint32_t n_intermediates,
                 uint64_t* out_handle) {
  if (n_dynamic > GE_MAX_DYN) return hm_fail(HM_ERR_BAD_ARG, "graph: more than 16 per-call constants");
  const size_t n_const = n_const_static + n_dynamic;
  if (n_calc > (1u << 24) || n_const >= (1u << 20) || n_rot > 1024 || n_columns > GE_MAX_COLUMNS || n_intermediates >= (1u << 20))
    return hm_fail(HM_ERR_BAD_ARG, "graph: program too large for the instruction encoding");
  auto g = std::make_unique<GraphProgram>();
  g->n_columns = n_columns;
  g->n_static = (uint32_t)n_const_static;
  g->n_dynamic = (uint32_t)n_dynamic;
  g->n_intermediates = n_intermediates;
  // validate
  std::vector<char> defined(n_intermediates, 0);
  for (size_t k = 0; k < n_calc; ++k) {
    const uint32_t* c = calcs5 + 5 * k;
    if (c[0] > GOP_MULADD) return hm_fail(HM_ERR_BAD_ARG, "graph: unknown operation");
    const int nsrc = c[0] == GOP_MULADD ? 3 : (c[0] <= GOP_MUL ? 2 : 1);
    for (int j = 0; j < nsrc; ++j) {
      const uint32_t s = c[1 + j];
      if (!src_ok(s, n_const, n_intermediates, n_columns, n_rot)) return hm_fail(HM_ERR_BAD_ARG, "graph: source out of range");
      if (gsrc_kind(s) == GSRC_INTER && !defined[gsrc_index(s)])
        return hm_fail(HM_ERR_BAD_ARG, "graph: intermediate read before it is written");
    }
    if (c[4] >= n_intermediates) return hm_fail(HM_ERR_BAD_ARG, "graph: target out of range");
    if (defined[c[4]]) return hm_fail(HM_ERR_BAD_ARG, "graph: intermediate written twice (every calculation owns its target)");
    defined[c[4]] = 1;
  }
  g->calcs5.assign(calcs5, calcs5 + 5 * n_calc);
  // constants -> internal form
  std::vector<uint32_t> c9(std::max<size_t>(n_const_static, 1) * 9, 0);
  for (size_t i = 0; i < n_const_static; ++i) host::fr_to_internal9(host::fr_load(constants_ext + 4 * i), &c9[9 * i]);
  const size_t b_const = c9.size() * 4, b_rot = std::max<size_t>(n_rot, 1) * 4;
  HM_HIP_CHECK(hipMalloc(&g->d_blob, b_const + b_rot));
  uint8_t* blob = (uint8_t*)g->d_blob;
  g->d_consts = (uint32_t*)blob;
  g->d_rot = (int32_t*)(blob + b_const);
  int rc = HM_OK;
  if (hipMemcpy(g->d_consts, c9.data(), b_const, hipMemcpyHostToDevice) != hipSuccess ||
      (n_rot && hipMemcpy(g->d_rot, rotations, n_rot * 4, hipMemcpyHostToDevice) != hipSuccess))
    rc = hm_fail(HM_ERR_HIP, "graph: upload failed");
  if (rc == HM_OK) rc = graph_lower(*g, false, g->variant[0]);    // the internal-columns form is lowered on first use
  if (rc != HM_OK) {
    graph_release(*g);
    return rc;
  }
  g->handle = ctx.next_handle++;
  *out_handle = g->handle;
  ctx.graphs.push_back(std::move(g));
  return HM_OK;
}

void graph_release(GraphProgram& g) {
  if (g.d_blob) (void)hipFree(g.d_blob);
  g.d_blob = nullptr;
  for (auto& v : g.variant) {
    if (v.d_calcs) (void)hipFree(v.d_calcs);
    v = GraphVariant{};
  }
}

// rows = segments << log_segment; rotations wrap inside a segment (segments == 1: the ordinary evaluation over 2^log_segment rows)
int graph_evaluate(DeviceCtx& ctx, GraphProgram& g, const void* const* d_columns, size_t n_columns, const uint64_t* dyn_ext,
                   size_t n_dyn, uint32_t log_size, uint32_t segments, void* d_values, uint32_t flags, hipStream_t stream) {
  if (flags & ~(uint32_t)HM_GRAPH_COLUMNS_INTERNAL) return hm_fail(HM_ERR_BAD_ARG, "graph: unknown flag");
  const bool internal_cols = (flags & HM_GRAPH_COLUMNS_INTERNAL) != 0;
  GraphVariant& v = g.variant[internal_cols ? 1 : 0];
  if (!v.ready) {
    const int rc = graph_lower(g, internal_cols, v);
    if (rc != HM_OK) return rc;
  }
  if (n_columns != g.n_columns) return hm_fail(HM_ERR_BAD_ARG, "graph: the program was built for another number of columns");
  if (n_dyn != g.n_dynamic) return hm_fail(HM_ERR_BAD_ARG, "graph: the program was built for another number of per-call constants");
  if (log_size > 30) return hm_fail(HM_ERR_BAD_ARG, "graph: log_size > 30");
  if (segments == 0 || ((uint64_t)segments << log_size) > (1ull << 32)) return hm_fail(HM_ERR_BAD_ARG, "graph: segments must be >= 1 and rows <= 2^32");
  const uint64_t size = (uint64_t)segments << log_size;
  // enough lanes to fill the chip, few enough that the intermediates' scratch stays cache-sized
  // 93 VGPRs: five waves per SIMD fit, i.e. five 256-lane workgroups per CU -- the interpreter's loads from the scratch
  // (Infinity Cache latency) need them: measured on the MerkleSumTree program over 2^21 rows, 512 / 768 / 1024 / 1280 / 2048
  // workgroups: 12.4 / 10.5 / 10.3 / 9.9 / 10.0 ms.  At 1280 the kernel issues 4.6e11 VALU wave-instructions per second
  // (SQ_INSTS_VALU 4.21e9 in 9.2 ms): it is bound by VALU issue like K3 and the NTT, no longer by its scratch traffic.
  static const uint32_t max_blocks = [] { const char* v = std::getenv("HALO2_MI355X_GRAPH_BLOCKS"); return (uint32_t)(v && *v ? std::atoi(v) : 1280); }();
  uint32_t blocks = (uint32_t)std::min<uint64_t>((size + GE_THREADS - 1) / GE_THREADS, max_blocks);
  const uint32_t T = blocks * GE_THREADS;
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  const size_t b_scratch = (size_t)v.n_slots * 9 * T * 4;
  uint8_t* buf = (uint8_t*)slot->scratch.ensure(b_scratch);
  if (!buf) return hm_fail(HM_ERR_HIP, "graph: scratch allocation failed");
  GraphColumns cols;
  std::memset(&cols, 0, sizeof cols);
  for (size_t i = 0; i < n_columns; ++i) {
    if (!d_columns[i]) return hm_fail(HM_ERR_BAD_ARG, "graph: null column pointer");
    cols.p[i] = (const uint32_t*)d_columns[i];
  }
  cols.n_static = g.n_static;
  for (size_t i = 0; i < n_dyn; ++i) host::fr_to_internal9(host::fr_load(dyn_ext + 4 * i), &cols.dyn[9 * i]);
  GraphColumns* d_cols = (GraphColumns*)slot->args.ensure(sizeof(GraphColumns));
  if (!d_cols) return hm_fail(HM_ERR_HIP, "graph: argument buffer allocation failed");
  // pageable source: the runtime has taken its copy of `cols` when this returns; the copy itself is ordered on `stream`
  // ahead of the launch and behind the previous launch that read the buffer
  HM_HIP_CHECK(hipMemcpyAsync(d_cols, &cols, sizeof cols, hipMemcpyHostToDevice, stream));
  if (internal_cols)
    hipLaunchKernelGGL(graph_evaluate_kernel<true>, dim3(blocks), dim3(GE_THREADS), 0, stream, (const GraphColumns*)d_cols,
                       (const uint32_t*)g.d_consts, (const int32_t*)g.d_rot, (const GraphCalc*)v.d_calcs, v.n_calc, v.result_src,
                       v.result_prev, (uint32_t*)buf, (uint32_t*)d_values, size, log_size);
  else
    hipLaunchKernelGGL(graph_evaluate_kernel<false>, dim3(blocks), dim3(GE_THREADS), 0, stream, (const GraphColumns*)d_cols,
                       (const uint32_t*)g.d_consts, (const int32_t*)g.d_rot, (const GraphCalc*)v.d_calcs, v.n_calc, v.result_src,
                       v.result_prev, (uint32_t*)buf, (uint32_t*)d_values, size, log_size);
  HM_HIP_CHECK(hipGetLastError());
  return aux_release(ctx, slot, stream);
}

}  // namespace hm
