// Varlen causal prefill attention for gfx950 (MI355X) on bf16 MFMA.
// Semantics of the reference call flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, ...,
// causal=True, block_table) at lightvllm/layers/attention.py:159-165: packed K/V, or K/V read from
// the paged cache through block_table when a prefix is cached; grouped-query heads; causal mask
// aligned to the bottom-right corner (query row r sees keys t <= r + Lk - Lq).
//
// Structure (flash-attention forward, one pass over K/V with online softmax):
//   * workgroup = 4 waves = one 128-row query tile of one (sequence, query head); wave w owns rows
//     32w..32w+31. Query tiles are launched heaviest (latest rows) first.
//   * K/V tiles of 64 keys are staged HBM -> VGPR -> LDS by all 256 lanes in whole 256-byte rows
//     (16 B per lane); the next tile's global loads are issued before the current tile's math so
//     their latency hides under it. The LDS image XOR-swizzles the 16-byte chunk index with the row
//     so that both the row reads (K as MFMA A operand) and the transposed reads (V^T as MFMA A
//     operand, ds_read_b64_tr_b16) are bank-conflict free.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: the accumulator then holds one query row per lane
//     (column) and 16 keys in registers, so row max / row sum are in-register plus one exchange
//     with lane^32, and the bf16-rounded P registers are directly the B operand of
//     O^T += V^T P^T (same lane = same query row), with no LDS round trip for P.
//   * fp32 scores, statistics and accumulation; exp2 with scale*log2(e) folded into one FMA.
// Two kernels share these ideas: prefill_attn_kernel (4 or 8 free-running waves, one barrier per
// tile) and prefill_pp_kernel (8 waves in two groups that alternate matrix and load clusters).
#include <stdlib.h>

#include <type_traits>

#include "lv_kernels.h"

namespace lv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) unsigned char lds_byte;  // LDS-typed pointers keep
// constant offsets in the ds_* instructions' immediate field instead of one address VGPR per read

constexpr int BN = 64;   // keys per staged tile
constexpr int kTileBytes = BN * kHeadDim * 2;  // 16 KiB

// Byte offset of 16-byte chunk `ch` (0..15) of row `row` inside a [rows][128 x bf16] LDS tile.
__device__ __forceinline__ int lds_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// Ablation builds (-DLV_ABL=mask, tools/prefill_diag.hip) drop components while keeping their
// inputs alive, to price them; results are then wrong by design. The shipped library has LV_ABL 0.
// bits: 1 exp, 2 PV MFMA, 4 QK MFMA, 8 global loads, 16 LDS staging writes, 32 K LDS reads,
// 64 V LDS reads, 128 softmax head (mask/max/rescale), 256 barriers, 512 output stores.
#ifndef LV_ABL
#define LV_ABL 0
#endif
#define LV_KEEP(x) asm volatile("" ::"v"(x))

// Diagnostic builds (-DLV_STAMPS, tools/prefill_diag.hip) accumulate s_memtime deltas per loop
// segment; the shipped library compiles these macros to nothing.
#ifdef LV_STAMPS
__device__ unsigned long long* g_lv_stamp_buf = nullptr;
#define LV_STAMP_DECL unsigned long long st_prev, st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
  __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory"); __builtin_amdgcn_sched_barrier(0);
#define LV_STAMP(i) { unsigned long long st_now; __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); __builtin_amdgcn_sched_barrier(0); \
  st_acc[i] += st_now - st_prev; st_prev = st_now; }
#define LV_STAMP_COUNT st_acc[15] += 1;
#define LV_CLOCK_BEGIN unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#define LV_CLOCK_END st_acc[12] = __builtin_amdgcn_s_memtime() - ck_t0; st_acc[13] = __builtin_amdgcn_s_memrealtime() - ck_r0;
#define LV_STAMP_FLUSH if (g_lv_stamp_buf && lane == 0) { \
  unsigned long long* d = g_lv_stamp_buf + static_cast<size_t>(blockIdx.x) * 128 + wave * 16; \
  for (int i = 0; i < 16; ++i) d[i] = st_acc[i]; }
#else
#define LV_STAMP_DECL
#define LV_STAMP(i)
#define LV_STAMP_COUNT
#define LV_STAMP_FLUSH
#define LV_CLOCK_BEGIN
#define LV_CLOCK_END
#endif

// Diagnostic builds (-DLV_TIMELINE, tools/prefill_diag.hip): wall-clock stamps (s_memrealtime, 100 MHz)
// at the seams of a workgroup's life in prefill_pp_kernel, kept in scalar registers and written once
// at the end together with the compute unit the workgroup ran on. Nothing in the shipped library.
#ifdef LV_TIMELINE
__device__ unsigned long long* g_lv_tl_buf = nullptr;
#define LV_TL_DECL unsigned long long tl_t[6] = {0, 0, 0, 0, 0, 0};
#define LV_TL(i) { __builtin_amdgcn_sched_barrier(0); tl_t[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
#define LV_TLA(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tl_now = __builtin_amdgcn_s_memrealtime(); \
  tl_t[i] += tl_now - tl_prev; tl_prev = tl_now; __builtin_amdgcn_sched_barrier(0); }
#define LV_TLA_START unsigned long long tl_prev = __builtin_amdgcn_s_memrealtime(); tl_t[2] = tl_t[3] = tl_t[4] = 0;
#define LV_TL_FLUSH if (g_lv_tl_buf && threadIdx.x == 0) { \
  unsigned long long* d = g_lv_tl_buf + static_cast<size_t>(blockIdx.x) * 8; \
  for (int i = 0; i < 6; ++i) d[i] = tl_t[i]; \
  d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
#else
#define LV_TL_DECL
#define LV_TL(i)
#define LV_TLA(i)
#define LV_TLA_START
#define LV_TL_FLUSH
#endif

// Rescale of the running output is skipped while no row's maximum grows by more than this many
// powers of two; P then reaches at most 2^kDeferLog2 instead of 1, which fp32 sums and the bf16
// P operand (same relative precision at any magnitude) absorb.
constexpr float kDeferLog2 = 6.0f;

template <int V>
struct IntC { static constexpr int value = V; };

// One 128-column output row from the lane pair (l, l ^ 32) that holds it: lane half h owns
// columns 8 j + 4 h .. + 3 of every 8-column group j (w[j], two bf16 per dword). One
// v_permlane32_swap per dword pairs groups 2k and 2k+1, after which the lower lane holds columns
// 16 k .. + 7 and the upper lane 16 k + 8 .. + 15 contiguously: 8 stores of 16 bytes per lane
// instead of 16 of 8 (the store tail of these kernels is bound by store instructions, not bytes).
// Every lane must call it (the exchange needs both halves); only valid rows store.
__device__ __forceinline__ void store_row_128(uint16_t* row_base, int h, bool valid, const u32x2 (&w)[16]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(w[2 * k][0], w[2 * k + 1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(w[2 * k][1], w[2 * k + 1][1], false, false);
    u32x4 v;
    v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
    if (valid) *reinterpret_cast<u32x4*>(row_base + 8 * h + 16 * k) = v;
  }
}

// max of x over the lane pair (l, l ^ 32), on the vector ALU (v_permlane32_swap, no LDS round trip).
// The two result words are copied to scalars before the bit cast: __builtin_bit_cast applied
// directly to a vector element (sw[1], w.y) compiles to element 0 with this hipcc.
__device__ __forceinline__ float pair32_max(float x) {
  const unsigned xb = __builtin_bit_cast(unsigned, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
  const unsigned lo = sw[0], hi = sw[1];
  return fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}

// ---- prefill_w64_kernel: registers owned by inline asm.
// At one wave per SIMD the wave has the whole 512-entry register file, but hipcc, left to place an
// accumulator that both MFMAs and vector code (the rare rescale) touch, moves it between the two
// halves of the file around every tile. So the output accumulator and the Q fragments are kept in
// accumulator registers that only asm statements name, and hipcc never sees them as values:
//   a[64:127]   O^T of row block 0 (column block dt at a[64 + 16 dt : +15])
//   a[128:191]  O^T of row block 1
//   a[192:223]  Q^T fragments of row block 0 (d-step s at a[192 + 4 s : +3]);  a[224:255] row block 1
// a[0:63] stay hipcc's: under register pressure it homes values there (the staging registers:
// global loads and LDS stores take accumulator registers directly). Nothing reserves a[64:255]
// for the asm statements, so after every edit the compiled kernel must be checked: no
// accumulator register above a63 outside ASMSTART/ASMEND (tests/test_abi_cpu.py does it).
// hipcc does not know these statements are MFMAs, so the wait states between an MFMA result and a
// vector instruction reading it, and between a vector write and an MFMA reading it, are placed by
// hand (s_nop inside or beside the statements).
#define LV_ACC_CLOBBERS_0_63 \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
  "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
  "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
#define LV_ACC_CLOBBERS_64_127 \
  "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79", \
  "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
  "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111", \
  "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
#define LV_ACC_CLOBBERS_128_191 \
  "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143", \
  "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159", \
  "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175", \
  "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"
#define LV_ACC_CLOBBERS_192_255 \
  "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207", \
  "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223", \
  "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239", \
  "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IntC<B>{});
    static_for<B + 1, E>(f);
  }
}

// scores (vector registers) = K fragment (vector) x Q^T fragment (accumulator registers QB..QB+3)
template <int QB, bool ZERO_C>
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& k) {
  if constexpr (ZERO_C)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(k), "n"(QB), "n"(QB + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(k), "n"(QB), "n"(QB + 3));
}
// O^T (accumulator registers OB..OB+15) += V^T fragment x P^T fragment (both vector registers)
template <int OB>
__device__ __forceinline__ void mfma_o(const bf16x8& v, const bf16x8& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(v), "v"(pf), "n"(OB), "n"(OB + 15));
}
template <int R>
__device__ __forceinline__ void acc_write(float x) {
  asm volatile("v_accvgpr_write_b32 a[%1:%1], %0" ::"v"(x), "n"(R));
}
template <int R>
__device__ __forceinline__ float acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1:%1]" : "=v"(x) : "n"(R));
  return x;
}
// a[R] *= alpha (vector register), through a scratch vector register
template <int R>
__device__ __forceinline__ void acc_scale(float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%2:%2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%2:%2], %0"
               : "=&v"(t) : "v"(alpha), "n"(R));
}

// Workgroup index -> its share of the launch: kv head, KV run, query head, sequence, first query row.
// Natural mapping (p.tile_order == nullptr): 1-D grid with the kv head fastest - workgroups are dealt
// round-robin over the 8 XCDs, so with 8 kv heads every query tile and query head that reads one
// (sequence, kv head)'s K/V runs on one XCD and shares its L2 - then KV run, query head within the
// group, query tile (heaviest = last rows first), sequence.
// With a caller-built order (the host knows the lengths): tile_order is a [rows][order_cols] table of
// items b * qtiles + t (sequence b, query tile t; negative = empty), order_cols = 8 / kv heads when
// that is 2, 4 or 8 and 1 otherwise. The workgroup index decomposes kv head fastest, then column,
// so XCD = kv head + kv heads * column: a column's items all run on one XCD per kv head. With fewer
// than 8 kv heads the natural mapping spreads every K/V stream over 8 / kv heads XCDs, each fetching
// it from HBM for itself; a table that keeps a sequence inside one or two columns lets an XCD's L2
// serve its stream once. Rows are taken in table order (heaviest first as the caller sorts them),
// and every row is followed by all query heads of its group.
struct TileWork { int kvh, split, hq, b, q0; };
template <int BM>
__device__ __forceinline__ bool tile_of_block(const PrefillParams& p, TileWork& w) {
  const int G = p.num_heads / p.num_kv_heads;
  const int qtiles = (p.max_seqlen_q + BM - 1) / BM;
  int wg = blockIdx.x;
  w.kvh = wg % p.num_kv_heads;
  wg /= p.num_kv_heads;
  if (p.tile_order == nullptr) {
    w.split = wg % p.nsplit;
    wg /= p.nsplit;
    w.hq = w.kvh * G + wg % G;
    wg /= G;
    w.q0 = (qtiles - 1 - wg % qtiles) * BM;
    w.b = wg / qtiles;
    return true;
  }
  const int col = wg % p.order_cols;
  wg /= p.order_cols;
  w.split = wg % p.nsplit;
  wg /= p.nsplit;
  w.hq = w.kvh * G + wg % G;
  wg /= G;
  const int item = p.tile_order[static_cast<int64_t>(wg) * p.order_cols + col];
  if (item < 0) return false;
  w.b = item / qtiles;
  w.q0 = (item % qtiles) * BM;
  return w.b < p.batch;
}

// NW = waves per workgroup, RB = 32-row query blocks per wave (1); the workgroup's query tile is
// 32*RB*NW rows, two waves per SIMD on compiler-allocated registers.
// DMA = true stages K/V tiles HBM -> LDS directly (global_load_lds, no VGPRs, no ds_write).
template <bool PAGED, int NW, int RB, bool DMA>
__global__ __launch_bounds__(64 * NW, (RB == 2 ? 1 : 2))
void prefill_attn_kernel(const PrefillParams p) {
  static_assert(RB == 1, "one 32-row block per wave (64 rows per wave: prefill_w64_kernel)");
  constexpr int BM = 32 * RB * NW;  // query rows per workgroup
  constexpr int NST = 16 / NW;      // 16-byte chunks of K (and of V) each thread stages per tile
  constexpr int RSTEP = 4 * NW;     // tile rows between two chunks of one thread
  // two buffers x (K tile | V tile)
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kTileBytes];
  lds_byte* const lds3 = (lds_byte*)lds;

  // With p.nsplit > 1 the key tiles of a query tile are cut into nsplit runs, one workgroup each
  // (few query tiles against a long context: a prefix-cached prompt, a short batch); the runs'
  // unnormalised outputs and (max, sum) go to the workspace and prefill_merge_kernel combines them.
  TileWork tw;
  if (!tile_of_block<BM>(p, tw)) return;
  const int kvh = tw.kvh, split = tw.split, hq = tw.hq, q0 = tw.q0, b = tw.b;
  const int q_start = p.cu_q[b];
  const int Lq = p.cu_q[b + 1] - q_start;
  const int k_start = p.cu_k[b];
  const int Lk = p.cu_k[b + 1] - k_start;
  if (q0 >= Lq) return;
  const int coff = Lk - Lq;  // causal offset: row r sees keys <= r + coff

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5;  // which 8-wide half of an MFMA k-step this lane feeds
  const int c = lane & 31;  // MFMA column = query row inside a 32-row block

  const int wrow0 = q0 + wave * (32 * RB);  // first row of the wave; block rb starts 32*rb later
  int my_lim[RB];                           // last visible key of this lane's row in block rb

  // Q^T fragments (B operand of S^T = K Q^T): per row block 8 k-steps x 8 bf16.
  bf16x8 qf[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int my_row = wrow0 + 32 * rb + c;
    my_lim[rb] = my_row + coff;
    const int row_ld = min(my_row, Lq - 1);
    const uint16_t* qp = p.q + static_cast<int64_t>(q_start + row_ld) * p.q_stride0 + hq * kHeadDim + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[rb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  // Retire the Q loads here, visibly to the compiler. Otherwise its wait-count bookkeeping carries
  // them as "possibly outstanding" into the tile loop and guards every use of a Q fragment with a
  // vmcnt wait that, in steady state, drains the K/V staging loads issued just before it.
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(qf[rb][s]));

  const int last_row = min(q0 + BM - 1, Lq - 1);
  const int max_key = min(last_row + coff, Lk - 1);
  const int ntiles_all = max_key >= 0 ? max_key / BN + 1 : 0;
  // this workgroup's run of key tiles: [t_lo, t_lo + ntiles)
  const int t_lo = static_cast<int>(static_cast<int64_t>(split) * ntiles_all / p.nsplit);
  const int ntiles = static_cast<int>(static_cast<int64_t>(split + 1) * ntiles_all / p.nsplit) - t_lo;

  // ---- staging: this thread moves chunk st_ch of tile rows st_r + RSTEP*i (i < NST).
  // Address = wave-uniform tile base (scalar registers) + a per-thread byte offset that is the
  // same for every tile, so the loop carries no 64-bit vector address arithmetic.
  const int st_ch = tid & 15;
  const int st_r = tid >> 4;
  const int32_t* bt_row = PAGED ? p.block_table + static_cast<int64_t>(b) * p.bt_stride : nullptr;
  const uint32_t k_row_bytes = static_cast<uint32_t>(p.k_stride0) * 2u;
  const uint32_t v_row_bytes = static_cast<uint32_t>(p.v_stride0) * 2u;
  const uint32_t col_bytes = static_cast<uint32_t>(kvh * kHeadDim + 8 * st_ch) * 2u;
  const uint32_t st_koff = static_cast<uint32_t>(st_r) * k_row_bytes + col_bytes;
  const uint32_t st_voff = static_cast<uint32_t>(st_r) * v_row_bytes + col_bytes;
  const int st_lds = lds_off(st_r, st_ch);  // rows st_r + RSTEP*i share the swizzle term: + 256*RSTEP*i

  u32x4 kreg[NST], vreg[NST];
  auto stage_load = [&](int tile) {
    if (LV_ABL & 8) return;
    const int kt0 = tile * BN;
    int64_t base_row;  // cache / packed row index of key kt0 (wave-uniform)
    if constexpr (PAGED) {
      const int blk = bt_row[kt0 / p.block_size];
      base_row = static_cast<int64_t>(blk) * p.block_size + (kt0 % p.block_size);
    } else {
      base_row = static_cast<int64_t>(k_start) + kt0;
    }
    const char* kb = reinterpret_cast<const char*>(p.k) + base_row * k_row_bytes;
    const char* vb = reinterpret_cast<const char*>(p.v) + base_row * v_row_bytes;
    if (kt0 + BN <= Lk) {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        kreg[i] = *reinterpret_cast<const u32x4*>(kb + static_cast<uint64_t>(RSTEP * i) * k_row_bytes + st_koff);
        vreg[i] = *reinterpret_cast<const u32x4*>(vb + static_cast<uint64_t>(RSTEP * i) * v_row_bytes + st_voff);
      }
    } else {
      // last, partial tile: rows past the sequence end re-read its last row (masked later)
      const int last = Lk - 1 - kt0;
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const uint32_t r = static_cast<uint32_t>(min(st_r + RSTEP * i, last));
        kreg[i] = *reinterpret_cast<const u32x4*>(kb + r * k_row_bytes + col_bytes);
        vreg[i] = *reinterpret_cast<const u32x4*>(vb + r * v_row_bytes + col_bytes);
      }
    }
  };
  auto stage_write = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    if (LV_ABL & 16) return;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      *reinterpret_cast<u32x4*>(lds + BUF * 2 * kTileBytes + 256 * RSTEP * i + st_lds) = kreg[i];
      *reinterpret_cast<u32x4*>(lds + BUF * 2 * kTileBytes + kTileBytes + 256 * RSTEP * i + st_lds) = vreg[i];
    }
  };
  // LDS-DMA staging: one wave instruction moves a 1-KiB piece = 4 tile rows. Wave w moves pieces
  // w + NW*i of K and of V. The LDS image of a piece is lane-linear (lane l -> row l>>4, physical
  // chunk l&15), so the swizzle is applied to the per-lane SOURCE address: physical chunk c of
  // row r must receive logical chunk c ^ x(r).
  auto stage_dma = [&](int tile, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    if (LV_ABL & 8) return;
    const int kt0 = tile * BN;
    int64_t base_row;
    if constexpr (PAGED) {
      const int blk = bt_row[kt0 / p.block_size];
      base_row = static_cast<int64_t>(blk) * p.block_size + (kt0 % p.block_size);
    } else {
      base_row = static_cast<int64_t>(k_start) + kt0;
    }
    const char* kb = reinterpret_cast<const char*>(p.k) + base_row * k_row_bytes;
    const char* vb = reinterpret_cast<const char*>(p.v) + base_row * v_row_bytes;
    const int last = Lk - 1 - kt0;
    const int lrow = lane >> 4;                                   // row inside the piece
    const int lchunk = (lane & 15) ^ ((lrow << 2) | (wave & 3));  // x(r): r&3 = lrow, (r>>2)&3 = wave&3
    const uint32_t colb = static_cast<uint32_t>(kvh * kHeadDim + 8 * lchunk) * 2u;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int piece = wave + NW * i;
      const uint32_t r = static_cast<uint32_t>(min(4 * piece + lrow, last));
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(kb + r * k_row_bytes + colb),
          (__attribute__((address_space(3))) void*)(lds3 + BUF * 2 * kTileBytes + 1024 * piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(vb + r * v_row_bytes + colb),
          (__attribute__((address_space(3))) void*)(lds3 + BUF * 2 * kTileBytes + kTileBytes + 1024 * piece), 16, 0, 0);
    }
  };

  f32x16 oacc[RB][4];
  float m_run[RB], l_run[RB];  // running row max (raw score units); this lane's share of the row sum
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    m_run[rb] = -INFINITY;
    l_run[rb] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[rb][dt][r] = 0.f;
  }

  const float cexp = p.scale_log2e;
  const float defer_raw = kDeferLog2 / cexp;  // threshold in raw score units

  // LDS read addresses inside buffer 0 (everything else is a compile-time offset):
  //   K as A operand: row c of a 32-key sub-tile, chunk (2s+h) swizzled with the row   -> 8 bases
  //   V^T as A operand (ds_read_b64_tr_b16, per 16-lane group): lane 4q+p of group gi addresses
  //   row 8*second + 4*(gi>>1) + q, columns 32*dt + 16*(gi&1) + 4p .. +3                 -> 8 bases
  const int xr = ((c & 3) << 2) | ((c >> 2) & 3);
  const int gi = lane >> 4;
  const int tq = (lane & 15) >> 2;
  const int tp = lane & 3;
  lds_byte* kaddr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) kaddr[s] = lds3 + 256 * c + 16 * ((2 * s + h) ^ xr);
  lds_byte* vaddr[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int second = 0; second < 2; ++second) {
      const int row = 8 * second + 4 * (gi >> 1) + tq;
      const int ch = 4 * dt + 2 * (gi & 1) + (tp >> 1);
      vaddr[dt][second] = lds3 + kTileBytes + lds_off(row, ch) + 8 * (tp & 1);
    }

  LV_STAMP_DECL

  // ---- pieces of one 32-key online-softmax step (row block 0; RB == 1)
  auto load_ka = [&](bf16x8 (&ka)[8], int byte_off) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
      ka[s] = (LV_ABL & 32) ? qf[0][s]
                            : *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kaddr[s] + byte_off);
  };
  auto load_va = [&](bf16x8 (&va)[2][4], int byte_off) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int second = 0; second < 2; ++second) {
          bf16x4 t;
          if (LV_ABL & 64) { t[0] = qf[0][dt][0]; t[1] = qf[0][dt][1]; t[2] = qf[0][s][2]; t[3] = qf[0][second][3]; }
          else t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(vaddr[dt][second] + byte_off + 4096 * s));
          va[s][dt][4 * second + 0] = t[0];
          va[s][dt][4 * second + 1] = t[1];
          va[s][dt][4 * second + 2] = t[2];
          va[s][dt][4 * second + 3] = t[3];
        }
  };
  // S^T = K Q^T
  auto qk = [&](const bf16x8 (&ka)[8], f32x16& sacc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (LV_ABL & 4) { LV_KEEP(ka[s]); continue; }
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s], qf[0][s], sacc, 0, 0, 0);
    }
  };
  // causal mask + row maximum of one step; returns the row maximum (all 64 lanes)
  auto row_max = [&](int ks0, f32x16& sacc) -> float {
    if (ks0 + 31 > wrow0 + coff) {  // only steps that reach past the wave's first row limit
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = ks0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (key > my_lim[0]) sacc[r] = -INFINITY;
      }
    }
    float mx = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    return pair32_max(mx);
  };
  // Running-max update. The accumulator is rescaled only when some row of the wave outgrows its
  // running maximum by more than the deferral threshold (always on a row's first visible keys,
  // m_run = -inf). Returns the factor for O and l (1 when nothing changes) through `alpha`.
  auto need_rescale = [&](float mx) -> bool {
    if (LV_ABL & 128) return false;
    return __builtin_amdgcn_ballot_w64(mx - m_run[0] > defer_raw) != 0;
  };
  auto new_max = [&](float mx) -> float {  // updates m_run, returns alpha
    const float m_new = fmaxf(m_run[0], mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m_run[0] - m_use) * cexp);
    m_run[0] = m_new;
    return alpha;
  };
  auto rescale = [&](float alpha) {
    l_run[0] *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[0][dt][r] *= alpha;
  };
  // P = exp2(S*c - m*c) against the current running max, bf16 P fragments (B operand of
  // O^T += V^T P^T); returns this lane's share of the row sum
  auto softmax_tail = [&](const f32x16& sacc, bf16x8 (&pf)[2]) -> float {
    const float mc = ((m_run[0] == -INFINITY) ? 0.f : m_run[0]) * cexp;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pe = (LV_ABL & 1) ? fmaf(sacc[r], cexp, -mc) : __builtin_amdgcn_exp2f(fmaf(sacc[r], cexp, -mc));
      psum += pe;
      pf[r >> 3][r & 7] = static_cast<__bf16>(pe);
    }
    return psum;
  };
  auto pv = [&](const bf16x8 (&va)[2][4], const bf16x8 (&pf)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if (LV_ABL & 2) { LV_KEEP(va[s][dt]); LV_KEEP(pf[s]); continue; }
        oacc[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[s][dt], pf[s], oacc[0][dt], 0, 0, 0);
      }
  };
  // Scheduling hint for a region holding 8 MFMAs and a block of vector work: one MFMA, then PER
  // vector instructions, eight times, so the vector work issues in the MFMAs' shadow.
  auto interleave8 = [&](auto per_c) {
    constexpr int PER = decltype(per_c)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);      // plain vector ALU
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);        // transcendental (v_exp_f32)
    }
  };

  // One 64-key tile = two 32-key online-softmax steps A and B, software pipelined inside the
  // wave so that matrix and vector work of different steps can issue together:
  //   QK(A) | max(A) | [QK(B) with exp(A)] | max(B) | [PV(A) with exp(B)] | rescale(B)? | PV(B)
  // A step's P is exponentiated against the running max in force when its max was taken; the
  // accumulator rescale that step B may ask for is applied after PV(A) has been added (PV(A) was
  // formed at the old scale) and before PV(B).
  auto compute = [&](int tile, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr int kBufOff = BUF * 2 * kTileBytes;
    const int kt0 = tile * BN;
    // waves whose rows all lie above this tile have nothing to add
    if (kt0 > wrow0 + 31 + coff) return;
    LV_STAMP_COUNT

    bf16x8 kx[8];       // K fragments of A, then of B
    bf16x8 va[2][4];    // V^T fragments of A
    bf16x8 vb[2][4];    // V^T fragments of B
    bf16x8 pa[2], pb[2];
    f32x16 sa, sb;

    load_ka(kx, kBufOff);
    qk(kx, sa);
    __builtin_amdgcn_sched_barrier(0);
    load_ka(kx, kBufOff + 8192);
    load_va(va, kBufOff);
    LV_STAMP(4)
    const float mxa = row_max(kt0, sa);
    if (need_rescale(mxa)) rescale(new_max(mxa));
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(8)
    // ---- region 1: S_B = K_B Q^T  with  P_A = exp(S_A)
    qk(kx, sb);
    const float psa = softmax_tail(sa, pa);
    // pin the vector work inside this region (pure arithmetic would otherwise sink to its users)
    asm volatile("" ::"v"(pa[0]), "v"(pa[1]), "v"(psa));
    interleave8(IntC<5>{});
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(9)
    l_run[0] += psa;
    load_va(vb, kBufOff + 8192);
    const float mxb = row_max(kt0 + 32, sb);
    const bool nb = need_rescale(mxb);
    float alpha_b = 1.f;
    if (nb) alpha_b = new_max(mxb);
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(10)
    // ---- region 2: O += V_A^T P_A  with  P_B = exp(S_B)
    pv(va, pa);
    const float psb = softmax_tail(sb, pb);
    asm volatile("" ::"v"(pb[0]), "v"(pb[1]), "v"(psb));
    interleave8(IntC<5>{});
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(5)
    if (nb) rescale(alpha_b);
    l_run[0] += psb;
    // ---- region 3: O += V_B^T P_B
    pv(vb, pb);
    LV_STAMP(6)
  };


  // ---- main loop, LDS-DMA staging: tile t+1 is copied into the other buffer while tile t is
  // computed; the barrier at the end of the interval (its fence waits for the copies) publishes it.
  if constexpr (DMA) {
    if (ntiles > 0) {
      stage_dma(t_lo, IntC<0>{});
      __syncthreads();
    }
    for (int tile = 0; tile < ntiles; tile += 2) {
      if (tile + 1 < ntiles) stage_dma(t_lo + tile + 1, IntC<1>{});
      LV_STAMP(3)
      compute(t_lo + tile, IntC<0>{});
      LV_STAMP(1)
      if (!(LV_ABL & 256)) __syncthreads();
      LV_STAMP(0)
      if (tile + 1 < ntiles) {
        if (tile + 2 < ntiles) stage_dma(t_lo + tile + 2, IntC<0>{});
        LV_STAMP(3)
        compute(t_lo + tile + 1, IntC<1>{});
        LV_STAMP(1)
        if (!(LV_ABL & 256)) __syncthreads();
        LV_STAMP(0)
      }
    }
  } else {
  // ---- main loop: tile t is computed from buffer t&1 while tile t+1 sits in registers and tile
    // t+2's loads are in flight; one workgroup barrier per tile.
    if (ntiles > 0) {
      stage_load(t_lo);
      stage_write(IntC<0>{});
      __syncthreads();
      if (ntiles > 1) stage_load(t_lo + 1);
    }
    for (int tile = 0; tile < ntiles; tile += 2) {
      compute(t_lo + tile, IntC<0>{});
      if (tile + 1 < ntiles) stage_write(IntC<1>{});
      LV_STAMP(1)
      // the next loads are issued before the barrier: a wave that gets here early spends its
      // wait issuing them, instead of all waves queueing on the memory pipe right after it
      if (tile + 2 < ntiles) stage_load(t_lo + tile + 2);
      LV_STAMP(3)
      if (!(LV_ABL & 256)) __syncthreads();
      LV_STAMP(0)
      if (tile + 1 < ntiles) {
        compute(t_lo + tile + 1, IntC<1>{});
        if (tile + 2 < ntiles) stage_write(IntC<0>{});
        LV_STAMP(1)
        if (tile + 3 < ntiles) stage_load(t_lo + tile + 3);
        LV_STAMP(3)
        if (!(LV_ABL & 256)) __syncthreads();
        LV_STAMP(0)
      }
    }
  }
  LV_STAMP_FLUSH

  // ---- epilogue: normalise and store this lane's rows (d = 32*dt + (r&3) + 8*(r>>2) + 4*h)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int my_row = wrow0 + 32 * rb + c;
    if (p.nsplit > 1) {
      // this run's share of the row: unnormalised output (this lane's 64 of the 128 columns, in
      // runs of four: d = 32 dt + 8 rg + 4 h + 0..3) and the (maximum, sum) pair the merge weighs it by
      if (my_row < Lq) {
        const int64_t slot = (static_cast<int64_t>(q_start + my_row) * p.num_heads + hq) * p.nsplit + split;
        float* wo = p.ws_o + slot * kHeadDim + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            *reinterpret_cast<float4*>(wo + 32 * dt + 8 * rg) =
                make_float4(oacc[rb][dt][4 * rg + 0], oacc[rb][dt][4 * rg + 1], oacc[rb][dt][4 * rg + 2], oacc[rb][dt][4 * rg + 3]);
        if (h == 0) {
          p.ws_ml[slot * 2 + 0] = m_run[rb] * cexp;    // -inf for a row that saw no key in this run
          p.ws_ml[slot * 2 + 1] = l_tot;
        }
      }
      continue;
    }
    u32x2 w[16];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        w[4 * dt + rg][0] = pack_bf16x2(oacc[rb][dt][4 * rg + 0] * inv, oacc[rb][dt][4 * rg + 1] * inv);
        w[4 * dt + rg][1] = pack_bf16x2(oacc[rb][dt][4 * rg + 2] * inv, oacc[rb][dt][4 * rg + 3] * inv);
      }
    uint16_t* ob = p.o + static_cast<int64_t>(q_start + min(my_row, Lq - 1)) * p.o_stride0 + hq * kHeadDim;
    store_row_128(ob, h, my_row < Lq && !(LV_ABL & 512), w);
  }
}


// ---------------------------------------------------------------------------------------------
// Phase-locked variant for long prompts: 8 waves x 32 rows, two wave groups (waves 0-3, 4-7; wave
// w and w+4 share a SIMD) that run the same four-cluster program per K/V tile one cluster apart,
// so each SIMD always has one wave in a matrix cluster beside its partner in a load cluster:
//
//   L_K(i)  read the K(i+1) and Q fragments (and apply a pending accumulator rescale, rare)
//   C_QK(i) S(i+1) = K(i+1) Q^T                                  (16 MFMAs, operands in registers)
//   L_V(i)  publish V(i+1) and K(i+2) to LDS, issue the global loads of V(i+2) and K(i+3), read
//           three quarters of the V(i) fragments (causal mask of tile i+1 on diagonal tiles)
//   C_PV(i) O += V(i)^T P(i) with the whole softmax of tile i+1 (row maximum, running-max update,
//           exponentials) issued between its MFMAs                                  (16 MFMAs)
//
// One workgroup barrier closes every cluster. Load clusters carry no vector ALU work: beside the
// partner's matrix cluster it would wait for issue slots. The P of tile i and the scores of tile
// i+1 are live together (two named register sets, alternating with the tile parity). A rescale decided for tile i+1 while P(i) is still
// to be added is applied to l at once and to O in the next L_K, after O += V(i)^T P(i).
#ifndef LV_PP_LPRIO
#define LV_PP_LPRIO 2
#endif
#ifndef LV_PP_SKIP
#define LV_PP_SKIP 1   // a wave leaves out the tiles that hold no key any of its rows may see (0: every wave walks every tile)
#endif
// PERSIST (launched when the caller passes an item table, PrefillParams::order_lanes > 0): one
// workgroup per compute unit walks a list of query tiles the host has balanced for it instead of
// ending after one. What a workgroup pays per tile shrinks to reading its Q fragments and storing its
// output: the next tile's Q rows and first two K tiles arrive by LDS-DMA while the current tile is
// computed (Q into the image behind the slots, which is free once the fragments are in registers; K
// into a second pair of K slots that the tiles use alternately), its first V tile is requested
// before the output stores, and its lengths come from the host's table entry (one scalar load
// instead of three dependent ones). Measured on 4 x 4096 tokens, 32q/8kv, non-persistent: 2.8 us
// from entry to Q fragments, 1.3 us to the first published K/V tiles, 1.4 us of store tail and 1.4 us
// until the next workgroup starts on the compute unit, against 63 us in the tile loop.
template <bool PAGED, bool PERSIST>
__global__ __launch_bounds__(512, 2)
void prefill_pp_kernel(const PrefillParams p) {
  constexpr int NW = 8;
  constexpr int BM = 32 * NW;
  constexpr int NST = 2;      // 16-byte chunks of K (and of V) each thread stages per tile
  constexpr int RSTEP = 32;   // tile rows between them
  // LDS: K slots s = 0, 1 at kKOff + s * kKStep, V slots at kVOff + s * kVStep, then a 64-KiB image of
  // the workgroup's Q tile (8 KiB per wave, that wave's 32 rows in the K tiles' swizzled row layout);
  // PERSIST: behind it a second pair of K slots (the K slots in use alternate between the pairs
  // from one query tile to the next: kring = 0 or kRingB is added to every K address)
#ifndef LV_PX
#define LV_PX 0      // diagnostic builds only: 1 = __syncthreads() barriers, 2 = compiler-visible LDS stores, 4 = slot layout of the one-tile kernel (with LV_PABL 3)
#endif
  constexpr bool kNewLayout = PERSIST && !(LV_PX & 4);
  constexpr int kKOff = 0, kKStep = kNewLayout ? kTileBytes : 2 * kTileBytes;
  constexpr int kVOff = kNewLayout ? 2 * kTileBytes : kTileBytes, kVStep = kNewLayout ? kTileBytes : 2 * kTileBytes;
  constexpr int kQImg = 4 * kTileBytes;
  constexpr int kRingB = kQImg + BM * kHeadDim * 2;
  constexpr int kLdsBytes = kRingB + (PERSIST ? 2 * kTileBytes : 0);
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kLdsBytes];
  lds_byte* const lds3 = (lds_byte*)lds;

  LV_TL_DECL
  LV_TL(0)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int lane = tid & 63;
  const int h = lane >> 5;
  const int c = lane & 31;

  // ---- what this workgroup serves. Per query tile: qrow0 = row of q / o of the tile's first row,
  // nrows = its valid rows (1..256), lim0 = last visible key of its first row (row r sees keys
  // <= lim0 + r), Lk keys starting at packed row kref (PAGED: of sequence kref's block table).
  // Workgroup barrier of the tile loop. PERSIST keeps LDS-DMA copies (the next query tile's rows) in
  // flight across these barriers; __syncthreads() would drain them (its fence waits for every
  // outstanding LDS write), so there the barrier is the bare instruction behind a wait for this
  // wave's own LDS stores and loads. The copies are waited for (vmcnt(0)) once, after the tile loop.
  auto wg_barrier = [&]() {
    if constexpr (PERSIST && !(LV_PX & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  // Lane id recomputed where a query tile begins or ends (PERSIST): values derived from it there are
  // then not carried, in registers the tile loop has none to spare of, across the loop (hipcc spills
  // them otherwise, and every reload waits for all outstanding loads, the DMA copies included).
  auto fresh_lane = [&]() -> int {
    if constexpr (PERSIST) {
      int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(l));
      return l;
    } else {
      return lane;
    }
  };
  // PERSIST stores outputs inside its loop over query tiles; for hipcc those stores might alias the
  // item table and the block table, so it would read both with vector loads (and wait for every
  // outstanding load, the DMA copies included, at each). Neither is written during the launch: read
  // through the constant address space they stay scalar loads.
  typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;
  typedef std::conditional_t<PERSIST, const_i32_ptr, const int32_t*> table_ptr;
  const table_ptr tile_table = (table_ptr)p.tile_order;
  const table_ptr block_table = (table_ptr)p.block_table;
  int kvh, hq;
  int qrow0, nrows, lim0, kref, Lk;
  [[maybe_unused]] int it_col = 0, it_lane = 0, it_lanes = 1, it_depth = 0, it_k = 0;
  auto item_ptr = [&](int k) -> table_ptr {     // PERSIST: table entry k of this workgroup's lane
    return tile_table + (static_cast<int64_t>(k * it_lanes + it_lane) * p.order_cols + it_col) * 8;
  };
  if constexpr (PERSIST) {
    const int G = p.num_heads / p.num_kv_heads;
    int wg = blockIdx.x;
    kvh = wg % p.num_kv_heads;
    wg /= p.num_kv_heads;
    it_col = wg % p.order_cols;
    wg /= p.order_cols;
    hq = kvh * G + wg % G;
    it_lane = wg / G;
    it_lanes = p.order_lanes;
    it_depth = static_cast<int>(p.order_len / (8 * static_cast<int64_t>(p.order_cols) * it_lanes));
    if (it_depth <= 0) return;
    const table_ptr e = item_ptr(0);
    qrow0 = e[0]; nrows = e[1]; lim0 = e[2]; kref = e[3]; Lk = e[4];
    if (nrows <= 0) return;
  } else {
    TileWork tw;
    if (!tile_of_block<BM>(p, tw)) return;   // launched with nsplit == 1
    kvh = tw.kvh; hq = tw.hq;
    const int q_start = p.cu_q[tw.b];
    const int Lq = p.cu_q[tw.b + 1] - q_start;
    const int k_start = p.cu_k[tw.b];
    Lk = p.cu_k[tw.b + 1] - k_start;
    if (tw.q0 >= Lq) return;
    qrow0 = q_start + tw.q0;
    nrows = min(BM, Lq - tw.q0);
    lim0 = tw.q0 + Lk - Lq;
    kref = PAGED ? tw.b : k_start;
  }
  // derived per query tile
  int wlim0, my_lim, ntiles, wtiles;
  table_ptr bt_row;
  auto derive = [&]() {
    wlim0 = lim0 + wave * 32;              // causal limit of the wave's first row
    my_lim = wlim0 + (fresh_lane() & 31);
    const int max_key = min(lim0 + nrows - 1, Lk - 1);
    ntiles = max_key >= 0 ? max_key / BN + 1 : 0;
    // key tiles in which some row of THIS wave sees a key: on the workgroup's last tiles (its
    // diagonal block) the waves that own its upper rows have nothing left to add
    wtiles = (LV_PP_SKIP && wlim0 + 31 >= 0) ? min(ntiles, (wlim0 + 31) / BN + 1) : ntiles;
    bt_row = PAGED ? block_table + static_cast<int64_t>(kref) * p.bt_stride : nullptr;
  };
  derive();

  // Q^T fragments (B operand of S^T = K Q^T), pre-multiplied by scale*log2(e) so scores come out
  // of the MFMA in log2 units
  const int xr = ((c & 3) << 2) | ((c >> 2) & 3);
  const int kbase0 = (256 * c) | (16 * (h ^ xr));
  bf16x8 qf[8];
  // softmax scale * log2(e) = q_pow * c2 with q_pow a power of two and 1 <= c2 < 2: Q is multiplied
  // by q_pow only, which is exact in bf16 (the full factor would round every element of Q to 8
  // bits again: on scores of magnitude ~30 that alone moved outputs by up to 0.12), and c2 is
  // applied in fp32 where the exponentials are taken. Scores, running maxima and the deferral
  // threshold are kept in units of 1 / c2.
  const float q_pow = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p.scale_log2e) & 0xff800000u);
  const float c2 = p.scale_log2e / q_pow;
  const float defer_s = kDeferLog2 / c2;

  // ---- staging (register path): chunk st_ch of tile rows st_r and st_r + 32
  const int st_ch = tid & 15;
  const int st_r = tid >> 4;
  const uint32_t k_row_bytes = static_cast<uint32_t>(p.k_stride0) * 2u;
  const uint32_t v_row_bytes = static_cast<uint32_t>(p.v_stride0) * 2u;
  const uint32_t col_bytes = static_cast<uint32_t>(kvh * kHeadDim + 8 * st_ch) * 2u;
  const int st_lds = lds_off(st_r, st_ch);
  int kring = 0;              // PERSIST: byte offset of the K slot pair in use (0 or kRingB)

  u32x4 kreg[NST], vreg[NST];
  auto tile_row_of = [&](table_ptr bt, int kr, int tile) -> int64_t {  // cache / packed row of a tile's first key
    const int kt0 = tile * BN;
    if constexpr (PAGED) {
      const int blk = bt[kt0 / p.block_size];
      return static_cast<int64_t>(blk) * p.block_size + (kt0 % p.block_size);
    } else {
      return static_cast<int64_t>(kr) + kt0;
    }
  };
  auto tile_row = [&](int tile) -> int64_t { return tile_row_of(bt_row, kref, tile); };
  // per-thread byte offset inside a tile (same for every tile): the tile base stays in scalar
  // registers and the loop carries no 64-bit vector address arithmetic
  const uint32_t st_koff = static_cast<uint32_t>(st_r) * k_row_bytes + col_bytes;
  const uint32_t st_voff = static_cast<uint32_t>(st_r) * v_row_bytes + col_bytes;
  // Buffer loads: the tile base and extent live in a scalar descriptor, the per-thread offset is
  // one 32-bit register, and rows past the sequence end read as zeros (they are masked later).
  auto load_rows = [&](int tile, const uint16_t* base, uint32_t row_bytes, uint32_t st_off, u32x4 (&reg)[NST]) {
    if (tile >= ntiles || (LV_ABL & 8)) return;
    const char* tb = reinterpret_cast<const char*>(base) + tile_row(tile) * row_bytes;
    const uint32_t rows = static_cast<uint32_t>(min(BN, Lk - tile * BN));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb), 0, rows * row_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NST; ++i)
      reg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, st_off, RSTEP * i * row_bytes, 0));
  };
  // the same for a tile of another query tile (PERSIST: the next one's first V tile)
  auto load_rows_at = [&](int64_t row, int rows_valid, const uint16_t* base, uint32_t row_bytes, uint32_t st_off,
                          u32x4 (&reg)[NST]) {
    const char* tb = reinterpret_cast<const char*>(base) + row * row_bytes;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb), 0, static_cast<uint32_t>(rows_valid) * row_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NST; ++i)
      reg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, st_off, RSTEP * i * row_bytes, 0));
  };
  // K/V rows from the staging registers into an LDS slot. PERSIST issues these stores as inline
  // asm: while LDS-DMA copies are outstanding hipcc puts a wait for ALL outstanding loads in front of
  // every LDS store it can see (the copy might be aimed at the same bytes; it is not: the copies go
  // to the Q image and the other K slot pair), and the tile loop lives on loads that stay in flight.
  // The wait the store does need, for the load that fills `reg`, hipcc still places (the register is
  // an input of the statement); wg_barrier() waits for the stores themselves.
  uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds3));   // LDS address of the array (0)
  auto lds_store = [&](int at, auto off_c, const u32x4& val) {   // 16 bytes to LDS byte `at` + compile-time offset
    constexpr int OFF = decltype(off_c)::value;
    [[maybe_unused]] const uint32_t addr = lds_base + static_cast<uint32_t>(at);
    if constexpr (PERSIST && !(LV_PX & 2)) {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory");
    } else {
      *reinterpret_cast<u32x4*>(lds + OFF + at) = val;
    }
  };
  auto write_rows = [&](auto off_c, const u32x4 (&reg)[NST]) {
    constexpr int OFF = decltype(off_c)::value;
    if (LV_ABL & 16) return;
    static_assert(NST == 2);
    lds_store(st_lds, IntC<OFF>{}, reg[0]);
    lds_store(st_lds, IntC<OFF + 256 * RSTEP>{}, reg[1]);
  };
  auto write_k = [&](auto off_c, const u32x4 (&reg)[NST]) {
    constexpr int OFF = decltype(off_c)::value;
    if (LV_ABL & 16) return;
    int at = st_lds;
    // st_lds + kring, formed where it is used: as a value of its own it would take one more of the
    // registers the tile loop has none to spare of
    if constexpr (PERSIST) asm volatile("v_add_u32 %0, %1, %2" : "=v"(at) : "v"(st_lds), "s"(kring));
    lds_store(at, IntC<OFF>{}, reg[0]);
    lds_store(at, IntC<OFF + 256 * RSTEP>{}, reg[1]);
  };
  // LDS-DMA of whole rows into the swizzled row image (one wave instruction = one 1-KiB piece = 4
  // rows; lane l brings the 16 bytes that belong at physical chunk l & 15 of row l >> 4 of the piece).
  // Q: rows of the query tile at q row `qr0` with `nr` valid rows, this wave's own 32 into its 8 KiB of
  // the image. K: a 64-key tile at cache / packed row `row` with `rows_valid` keys into LDS offset
  // `dst`, pieces wave and wave + 8. Rows past the end re-read the last valid row (masked later).
  auto dma_q = [&](int qr0, int nr) {
    const int ln = fresh_lane();
    const int lrow = ln >> 4;
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
      const int r = 4 * pc + lrow;                               // row of the wave's 32
      const int lchunk = (ln & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
      const int row_ld = min(wave * 32 + r, nr - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(p.q + static_cast<int64_t>(qr0 + row_ld) * p.q_stride0 +
                                                          hq * kHeadDim + 8 * lchunk),
          (__attribute__((address_space(3))) void*)(lds3 + kQImg + wave * 8192 + 1024 * pc), 16, 0, 0);
    }
  };
  [[maybe_unused]] auto dma_k = [&](int64_t row, int rows_valid, int dst) {
    const int ln = fresh_lane();
    const int lrow = ln >> 4;
    const int lchunk = (ln & 15) ^ ((lrow << 2) | (wave & 3));   // pieces wave, wave + 8: (piece & 3) = wave & 3
    const char* kb = reinterpret_cast<const char*>(p.k) + row * k_row_bytes +
                     static_cast<uint32_t>(kvh * kHeadDim + 8 * lchunk) * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave + 8 * i;
      const uint32_t r = static_cast<uint32_t>(max(min(4 * piece + lrow, rows_valid - 1), 0));   // no keys: the tensor's row `row`
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(kb + r * k_row_bytes),
          (__attribute__((address_space(3))) void*)(lds3 + dst + 1024 * piece), 16, 0, 0);
    }
  };

  f32x16 oacc[4];
  float m_run, l_run, defer_thr, pend_alpha;
  bool pend;
  auto reset_state = [&]() {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    m_run = -INFINITY;
    l_run = 0.f;
    defer_thr = -INFINITY;
    pend_alpha = 1.f;
    pend = false;
  };
  reset_state();

  // LDS read addresses inside slot 0. The swizzled fragment addresses of prefill_attn_kernel
  // differ from one per-lane base only by XOR with compile-time constants:
  //   K, d-step s:                 kbase ^ 32 s
  //   V^T, column block dt, rows +8 ("second"):   vbase ^ 64 dt ^ (second ? 2048 | 32 : 0)
  // so two registers stand in for sixteen; the XORs are redone where the reads are issued.
  const int gi = lane >> 4;
  const int tq = (lane & 15) >> 2;
  const int tp = lane & 3;
  const int vbase0 = lds_off(4 * (gi >> 1) + tq, 2 * (gi & 1) + (tp >> 1)) + 8 * (tp & 1);   // inside a V slot

  // K fragments of four d-steps (s0 .. s0+3) of a 32-key half
  auto load_kg = [&](bf16x8 (&kg)[4], int kbase, int byte_off, auto s0_c) {
    constexpr int S0 = decltype(s0_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (LV_ABL & 32) { asm volatile("" : "=v"(kg[j])); continue; }
      kg[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(lds3 + (kbase ^ (32 * (S0 + j))) + byte_off);
    }
  };
  // V^T fragments of one 32-column block dt for the tile's four 16-key steps (index 2*half + s)
  auto load_vg = [&](bf16x8 (&vg)[4], int vbase, int byte_off, auto dt_c) {
    constexpr int DT = decltype(dt_c)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int second = 0; second < 2; ++second) {
        if (LV_ABL & 64) { if (second == 0) asm volatile("" : "=v"(vg[ks])); continue; }
        const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(lds3 + (vbase ^ (64 * DT) ^ (second * (2048 | 32))) + byte_off + 4096 * ks));
        vg[ks][4 * second + 0] = t[0];
        vg[ks][4 * second + 1] = t[1];
        vg[ks][4 * second + 2] = t[2];
        vg[ks][4 * second + 3] = t[3];
      }
  };
  auto qk4 = [&](const bf16x8 (&kg)[4], auto s0_c, const f32x16& cin, f32x16& sacc) {
    constexpr int S0 = decltype(s0_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (LV_ABL & 4) { LV_KEEP(kg[j]); continue; }
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kg[j], qf[S0 + j], j == 0 ? cin : sacc, 0, 0, 0);
    }
  };
  auto pv4 = [&](const bf16x8 (&vg)[4], const bf16x8 (&pf)[2][2], auto dt_c) {
    constexpr int DT = decltype(dt_c)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (LV_ABL & 2) { LV_KEEP(vg[ks]); LV_KEEP(pf[ks >> 1][ks & 1]); continue; }
      oacc[DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vg[ks], pf[ks >> 1][ks & 1], oacc[DT], 0, 0, 0);
    }
  };
  // scheduling hint: N x {1 MFMA, 3 vector ALU, 1 transcendental}
  // max(a, b, c) in one instruction: fmaxf on MFMA outputs compiles to a canonicalising
  // v_max per operand plus the max itself
  auto max3 = [](float a, float b, float c) -> float {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  };
  // causal mask of a tile that reaches the diagonal (uniform branch; runs in a load cluster)
  auto mask_tile = [&](int tile, f32x16& sa, f32x16& sb) {
    const int kt0 = tile * BN;
    if (kt0 + 63 > wlim0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (key > my_lim) sa[r] = -INFINITY;
        if (key + 32 > my_lim) sb[r] = -INFINITY;
      }
    }
  };
  // The MFMA accumulator of S(i+1) is seeded with -m_off, the running maximum after tile i (0
  // while a row has seen no key), so scores arrive as s - m_off in log2 units and P = exp2 of
  // them with no further arithmetic, as long as the maximum does not move. head() takes the row
  // maximum of such a tile; when some row of the wave outgrows m_off by more than the deferral
  // threshold (or has no maximum yet) it moves the maxima, shifts the tile's scores by the
  // difference, rescales l at once and leaves the O rescale pending for the next L_K, after the
  // P that is still at the old scale has been added.
  auto head = [&](f32x16& sa, f32x16& sb) {
    if (LV_ABL & 128) { LV_KEEP(sa); LV_KEEP(sb); m_run = 0.f; return; }
    float a[10];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      a[j] = max3(sa[3 * j], sa[3 * j + 1], sa[3 * j + 2]);
      a[5 + j] = max3(sb[3 * j], sb[3 * j + 1], sb[3 * j + 2]);
    }
    const float b0 = max3(a[0], a[1], a[2]);
    const float b1 = max3(a[3], a[4], a[5]);
    const float b2 = max3(a[6], a[7], a[8]);
    const float b3 = max3(a[9], sa[15], sb[15]);
    // plain v_max_f32 / v_max3_f32 on these values: fmaxf adds a canonicalising v_max per operand
    // to a chain that every exponential of the tile waits for
    float mx = max3(max3(b0, b1, b2), b3, b3);
    if (LV_ABL & 1024) { asm volatile("" ::"v"(mx)); m_run = 0.f; return; }
    {
      const unsigned xb = __builtin_bit_cast(unsigned, mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
      const unsigned lo = sw[0], hi = sw[1];
      const float flo = __builtin_bit_cast(float, lo), fhi = __builtin_bit_cast(float, hi);
      mx = max3(flo, fhi, fhi);
    }
    if (LV_ABL & 2048) { asm volatile("" ::"v"(mx)); m_run = 0.f; return; }
    // defer_thr is the deferral threshold for a row that has a maximum and -inf for one that has
    // none yet (any finite score then moves it): one compare decides both cases
    if (__builtin_amdgcn_ballot_w64(mx > defer_thr) != 0) {
      const bool fresh = m_run == -INFINITY;
      const float m_off = fresh ? 0.f : m_run;
      const float m_new = fmaxf(m_run, m_off + mx);
      const float delta = (m_new == -INFINITY) ? 0.f : m_new - m_off;
      const float alpha = fresh ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * c2);
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] -= delta; sb[r] -= delta; }
      l_run *= alpha;
      m_run = m_new;
      defer_thr = (m_new == -INFINITY) ? -INFINITY : defer_s;
      pend_alpha = alpha;
      pend = true;
    }
  };
  // a quarter of a 32-key half's exp: P = exp2(S) for registers 4q .. 4q+3
  auto tail_q = [&](const f32x16& sacc, bf16x8 (&pf)[2], float (&ps4)[4], auto q_c) {
    float& psum = ps4[decltype(q_c)::value];
    constexpr int Q = decltype(q_c)::value;
    if (LV_ABL & 512) { if (Q == 0) { asm volatile("" : "=v"(pf[0])); asm volatile("" : "=v"(pf[1])); } return; }
#pragma unroll
    for (int r = 4 * Q; r < 4 * Q + 4; ++r) {
      const float pe = (LV_ABL & 1) ? sacc[r] : __builtin_amdgcn_exp2f(sacc[r] * c2);
      psum += pe;
      pf[r >> 3][r & 7] = static_cast<__bf16>(pe);
    }
  };
  // scheduling hint for a slice of 4 MFMAs beside NV plain and NT transcendental vector
  // instructions per MFMA
  auto mix4 = [&](auto nv_c, auto nt_c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, decltype(nv_c)::value, 0);
      if (decltype(nt_c)::value > 0) __builtin_amdgcn_sched_group_barrier(0x400, decltype(nt_c)::value, 0);
    }
  };

  LV_STAMP_DECL
  f32x16 S[2][2];      // [tile parity][32-key half] scores
  bf16x8 P[2][2][2];   // [tile parity][32-key half][k-step] probabilities

  // One tile interval. PAR = parity of tile i; CUR: tile i exists (false only for i = -1).
  auto body = [&](int i, auto par_c, auto cur_c) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int NPAR = PAR ^ 1;
    constexpr bool CUR = decltype(cur_c)::value != 0;
    int kbase = kbase0 + kring, vbase = vbase0;   // opaque copies: keeps the XORs next to the reads
    asm volatile("" : "+v"(kbase), "+v"(vbase));
    LV_STAMP_COUNT
    bf16x8 kx[4][4];   // K fragments: [32-key half * 2 + d-step group][4]
    bf16x8 vx[3][4];   // V^T fragments: [column block][16-key step]; block 3 reuses block 0's registers
    // ---- L_K(i): loads only (vector ALU work here would compete with the partner's C cluster)
    __builtin_amdgcn_s_setprio(LV_PP_LPRIO);
    load_kg(kx[0], kbase, kKOff + NPAR * kKStep, IntC<0>{});
    load_kg(kx[1], kbase, kKOff + NPAR * kKStep, IntC<4>{});
    load_kg(kx[2], kbase, kKOff + NPAR * kKStep + 8192, IntC<0>{});
    load_kg(kx[3], kbase, kKOff + NPAR * kKStep + 8192, IntC<4>{});
    // K(i+2) replaces K(i), last read in L_K(i-1); its rows were requested in L_K(i-1)
    if constexpr (PERSIST && !CUR) {
      // K(0) and K(1) came by LDS-DMA while the previous query tile was computed and K(2) was
      // requested before this interval: nothing to publish or request here
    } else if (i + 2 < ntiles) {
      write_k(IntC<kKOff + PAR * kKStep>{}, kreg);
      __builtin_amdgcn_sched_barrier(0);
      load_rows(i + 3, p.k, k_row_bytes, st_koff, kreg);
    }
    if constexpr (CUR) {
      if (pend) {   // rare: O += V(i-1)^T P(i-1) is complete, P(i) is already at the new scale
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= pend_alpha;
        pend = false;
      }
    }
    // one splat of -m feeds the first MFMA of both 32-key halves as its C operand; formed here,
    // where the wave would otherwise only wait at the barrier, not at the head of the matrix cluster
    f32x16 seedv;
    {
      const float seed = (m_run == -INFINITY) ? 0.f : -m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) seedv[r] = seed;
      asm volatile("" : "+v"(seedv));
    }
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(0)
    if (!(LV_ABL & 256)) wg_barrier();
    LV_STAMP(4)
    __builtin_amdgcn_sched_barrier(0);
    // ---- C_QK(i): S(i+1) = K(i+1) Q^T, operands in registers
    __builtin_amdgcn_s_setprio(0);
    {
      float ps4[4] = {0.f, 0.f, 0.f, 0.f};
      qk4(kx[0], IntC<0>{}, seedv, S[NPAR][0]);
      if constexpr (CUR) tail_q(S[PAR][1], P[PAR][1], ps4, IntC<0>{});
      if constexpr (CUR) mix4(IntC<2>{}, IntC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      qk4(kx[1], IntC<4>{}, S[NPAR][0], S[NPAR][0]);
      if constexpr (CUR) tail_q(S[PAR][1], P[PAR][1], ps4, IntC<1>{});
      if constexpr (CUR) mix4(IntC<2>{}, IntC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      qk4(kx[2], IntC<0>{}, seedv, S[NPAR][1]);
      if constexpr (CUR) tail_q(S[PAR][1], P[PAR][1], ps4, IntC<2>{});
      if constexpr (CUR) mix4(IntC<2>{}, IntC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      qk4(kx[3], IntC<4>{}, S[NPAR][1], S[NPAR][1]);
      if constexpr (CUR) tail_q(S[PAR][1], P[PAR][1], ps4, IntC<3>{});
      if constexpr (CUR) mix4(IntC<2>{}, IntC<1>{});
      if constexpr (CUR) {
        l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
        asm volatile("" ::"v"(l_run), "v"(P[PAR][1][0]), "v"(P[PAR][1][1]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(1)
    if (!(LV_ABL & 256)) wg_barrier();
    LV_STAMP(5)
    __builtin_amdgcn_sched_barrier(0);
    // ---- L_V(i)
    __builtin_amdgcn_s_setprio(LV_PP_LPRIO);
    // V(i+1) replaces V(i-1), last read in C_PV(i-1), one cluster before the second group gets
    // here; its rows were requested in L_V(i-1)
    if (i + 1 < ntiles) {
      write_rows(IntC<kVOff + NPAR * kVStep>{}, vreg);
      __builtin_amdgcn_sched_barrier(0);
      load_rows(i + 2, p.v, v_row_bytes, st_voff, vreg);
    }
    LV_STAMP(8)
    if constexpr (CUR) {
      load_vg(vx[0], vbase, kVOff + PAR * kVStep, IntC<0>{});
      load_vg(vx[1], vbase, kVOff + PAR * kVStep, IntC<1>{});
      load_vg(vx[2], vbase, kVOff + PAR * kVStep, IntC<2>{});
    }
    LV_STAMP(9)
    mask_tile(i + 1, S[NPAR][0], S[NPAR][1]);
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(2)
    if (!(LV_ABL & 256)) wg_barrier();
    LV_STAMP(6)
    __builtin_amdgcn_sched_barrier(0);
    // ---- C_PV(i): O += V(i)^T P(i); in the MFMAs' shadow the first half of the softmax of tile
    // i+1: slice 0 the row maximum and running-max update, slices 1-2 the exponentials of keys
    // 0-31 (those of keys 32-63 run beside C_QK(i+1): 16 per matrix cluster either way; 12/20 and
    // 24/8 splits both measured slower)
    __builtin_amdgcn_s_setprio(0);
    {
      float ps4[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (CUR) pv4(vx[0], P[PAR], IntC<0>{});
      head(S[NPAR][0], S[NPAR][1]);
      if constexpr (CUR) mix4(IntC<6>{}, IntC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CUR) { if (!(LV_ABL & 8192)) load_vg(vx[0], vbase, kVOff + PAR * kVStep, IntC<3>{}); }
      if constexpr (CUR) pv4(vx[1], P[PAR], IntC<1>{});
      tail_q(S[NPAR][0], P[NPAR][0], ps4, IntC<0>{});
      tail_q(S[NPAR][0], P[NPAR][0], ps4, IntC<1>{});
      asm volatile("" ::"v"(ps4[0]), "v"(ps4[1]));
      if constexpr (CUR) mix4(IntC<3>{}, IntC<2>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CUR) pv4(vx[2], P[PAR], IntC<2>{});
      tail_q(S[NPAR][0], P[NPAR][0], ps4, IntC<2>{});
      tail_q(S[NPAR][0], P[NPAR][0], ps4, IntC<3>{});
      asm volatile("" ::"v"(ps4[2]), "v"(ps4[3]), "v"(P[NPAR][0][0]), "v"(P[NPAR][0][1]));
      if constexpr (CUR) mix4(IntC<3>{}, IntC<2>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CUR) pv4(vx[0], P[PAR], IntC<3>{});
      l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
      asm volatile("" ::"v"(l_run));
    }
    __builtin_amdgcn_sched_barrier(0);
    LV_STAMP(3)
    if (!(LV_ABL & 256)) wg_barrier();
    LV_STAMP(7)
    __builtin_amdgcn_sched_barrier(0);
  };

  // One tile interval of a wave that has nothing to compute any more: its share of the staging, and
  // the interval's four barriers.
  auto stage_only = [&](int i, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int NPAR = PAR ^ 1;
    __builtin_amdgcn_s_setprio(LV_PP_LPRIO);
    if (i + 2 < ntiles) {
      write_k(IntC<kKOff + PAR * kKStep>{}, kreg);
      __builtin_amdgcn_sched_barrier(0);
      load_rows(i + 3, p.k, k_row_bytes, st_koff, kreg);
    }
    __builtin_amdgcn_sched_barrier(0);
    wg_barrier();
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (i + 1 < ntiles) {
      write_rows(IntC<kVOff + NPAR * kVStep>{}, vreg);
      __builtin_amdgcn_sched_barrier(0);
      load_rows(i + 2, p.v, v_row_bytes, st_voff, vreg);
    }
    __builtin_amdgcn_sched_barrier(0);
    wg_barrier();
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // Q fragments out of the wave's Q image (landed and waited for by this wave)
  auto read_q = [&]() {
    bf16x8 qtmp[8];
    const int ln = fresh_lane();
    const int cq = ln & 31;
    const int kb = (256 * cq) | (16 * ((ln >> 5) ^ (((cq & 3) << 2) | ((cq >> 2) & 3))));    // = kbase0
#pragma unroll
    for (int s = 0; s < 8; ++s)
      qtmp[s] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(lds3 + kQImg + wave * 8192 + (kb ^ (32 * s)));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = static_cast<__bf16>(static_cast<float>(qtmp[s][j]) * q_pow);
      asm volatile("" ::"v"(qf[s]));   // retire the reads before the tile loop
    }
  };
  // the tile loop of one query tile: K(0) (PERSIST: and K(1)) published, V(0) in vreg
  auto tile_loop = [&]() {
    // K(0) published, K(1) and V(0) in flight (L_K(-1) publishes K(1), L_V(-1) V(0))
    LV_CLOCK_BEGIN
    body(-1, IntC<1>{}, IntC<0>{});
    // Every interval also forms the scores of tile i+1; past the wave's last tile they are computed
    // from stale LDS and fully masked (every key of that tile lies beyond every row's causal limit),
    // which leaves the running statistics untouched and keeps the loop body free of branches
    // around its matrix clusters. (An interval variant without that next tile, for the wave's last
    // one, measured no faster and spilled.)
    // A wave walks only the tiles that hold a key some row of it may see (wtiles of the workgroup's
    // ntiles: on the diagonal block the waves of the upper rows are done early). For the rest it
    // keeps its share of the staging and the four barriers of every interval, but reads no
    // fragments, issues no MFMAs and takes no exponentials: those tiles' probabilities were exact
    // zeros, so the result is the same to the bit, and the chip, which runs this kernel at its power
    // limit, is spared the work (4 x 4096 tokens, 32q/8kv, same box, alternating runs: 0.566 -> 0.548 ms).
    for (int i = 0; i < wtiles; i += 2) {
      body(i, IntC<0>{}, IntC<1>{});
      if (i + 1 < wtiles) body(i + 1, IntC<1>{}, IntC<1>{});
    }
    for (int i = wtiles; i < ntiles; ++i) {
      if (i & 1) stage_only(i, IntC<1>{});
      else stage_only(i, IntC<0>{});
    }
    if (grp == 0) wg_barrier();
    LV_CLOCK_END
  };
  auto store_out = [&]() {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    u32x2 w[16];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        w[4 * dt + rg][0] = pack_bf16x2(oacc[dt][4 * rg + 0] * inv, oacc[dt][4 * rg + 1] * inv);
        w[4 * dt + rg][1] = pack_bf16x2(oacc[dt][4 * rg + 2] * inv, oacc[dt][4 * rg + 3] * inv);
      }
    const int ln = fresh_lane();
    const int row = wave * 32 + (ln & 31);
    uint16_t* ob = p.o + static_cast<int64_t>(qrow0 + min(row, nrows - 1)) * p.o_stride0 + hq * kHeadDim;
    store_row_128(ob, ln >> 5, row < nrows, w);
  };

  if constexpr (!PERSIST) {
    // ---- prologue. The first K/V tiles are requested before Q, and Q comes in whole 256-byte rows
    // by LDS-DMA (one wave instruction = 4 rows of the wave's own 32, so only the wave's own wait
    // stands between the copy and its fragment reads): the three latencies overlap, and no lane
    // gathers 32-byte pieces of 32 different rows.
    load_rows(0, p.k, k_row_bytes, st_koff, kreg);
    load_rows(0, p.v, v_row_bytes, st_voff, vreg);
    dma_q(qrow0, nrows);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ntiles > 0) write_k(IntC<kKOff>{}, kreg);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(1, p.k, k_row_bytes, st_koff, kreg);
    read_q();
    LV_TL(1)
    if (ntiles > 0) {
      __syncthreads();
      if (grp == 1) __syncthreads();   // the second group runs one cluster behind the first
      LV_TL(2)
      tile_loop();
      LV_TL(3)
    }
    LV_STAMP_FLUSH
    store_out();
    LV_TL(4)
#ifdef LV_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave's queue
    LV_TL(5)
#endif
    LV_TL_FLUSH
  } else {
    // ---- cold start: this workgroup's first query tile arrives the way every later one does, only
    // nothing is computed meanwhile
    dma_q(qrow0, nrows);
    if (Lk > 0) {
      dma_k(tile_row(0), min(BN, Lk), kKOff);
      dma_k(tile_row(Lk > BN ? 1 : 0), Lk > BN ? min(BN, Lk - BN) : min(BN, Lk), kKOff + kKStep);
      load_rows_at(tile_row(0), min(BN, Lk), p.v, v_row_bytes, st_voff, vreg);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the builtin, so that hipcc counts from here
    LV_TL(1)
    LV_TLA_START      // diagnostic builds: tl_t[2] += seam (Q fragments, barrier), [3] += tile loop, [4] += wait + output stores
    while (true) {
      // entry of the table for the query tile after this one (nrows <= 0: none); a scalar load that
      // lands while the first cluster of this tile runs
      int n_qrow0 = 0, n_nrows = 0, n_lim0 = 0, n_kref = 0, n_Lk = 0;
      if (it_k + 1 < it_depth) {
        const table_ptr e = item_ptr(it_k + 1);
        n_qrow0 = e[0]; n_nrows = e[1]; n_lim0 = e[2]; n_kref = e[3]; n_Lk = e[4];
      }
      read_q();
      reset_state();
      // the DMA'd K tiles of every wave are in LDS (each wave waited for its own after the last tile
      // loop) and the old tile's LDS reads are over
      wg_barrier();
      if (ntiles > 0 && grp == 1) wg_barrier();   // the second group runs one cluster behind the first
      LV_TLA(2)
      load_rows(2, p.k, k_row_bytes, st_koff, kreg);   // K(0), K(1) are in LDS already; before the copies below, so
                                                       // that the wait for K(2) does not wait for them as well
      // ---- the tile after this one: Q rows into the image (this wave's fragments are in registers),
      // first two K tiles into the other K slot pair. Without a next tile the same copies are made
      // of this tile's own rows (into places nobody reads any more): the instruction stream, and with
      // it the compiler's count of outstanding loads, is the same for every tile.
      const bool more = n_nrows > 0;
      {
        const int f_qrow0 = more ? n_qrow0 : qrow0, f_nrows = more ? n_nrows : nrows;
        const int f_kref = more ? n_kref : kref, f_Lk = more ? n_Lk : Lk;
        const table_ptr f_bt = PAGED ? block_table + static_cast<int64_t>(f_kref) * p.bt_stride : nullptr;
#ifndef LV_PABL
#define LV_PABL 0     // diagnostic builds only: 1 = no copies for the next query tile (results then wrong)
#endif
        if (!(LV_PABL & 1)) dma_q(f_qrow0, f_nrows);
        // (no branch around the copies: a conditional copy would leave hipcc unable to count the
        // outstanding loads behind it, and its next wait would be for all of them)
        const int other = kring ^ kRingB;
        const bool second = f_Lk > BN;
        if (!(LV_PABL & 1)) {
          dma_k(f_Lk > 0 ? tile_row_of(f_bt, f_kref, 0) : 0, min(BN, f_Lk), other + kKOff);
          dma_k(f_Lk > 0 ? tile_row_of(f_bt, f_kref, second ? 1 : 0) : 0, second ? min(BN, f_Lk - BN) : min(BN, f_Lk),
                other + kKOff + kKStep);
        }
      }
      if (ntiles > 0) tile_loop();
      LV_TLA(3)
      // every copy for the next tile has landed (they are older than every load of the loop, so this
      // wait is short); its first V tile is requested before this tile's output is stored
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the builtin, so that hipcc counts from here
      if (more && n_Lk > 0 && !(LV_PABL & 2)) {
        const table_ptr n_bt = PAGED ? block_table + static_cast<int64_t>(n_kref) * p.bt_stride : nullptr;
        load_rows_at(tile_row_of(n_bt, n_kref, 0), min(BN, n_Lk), p.v, v_row_bytes, st_voff, vreg);
      }
      store_out();
      LV_TLA(4)
      if (!more) break;
      qrow0 = n_qrow0; nrows = n_nrows; lim0 = n_lim0; kref = n_kref; Lk = n_Lk;
      derive();
      if (!(LV_PX & 4)) kring ^= kRingB;
      ++it_k;
    }
    LV_STAMP_FLUSH
#ifdef LV_TIMELINE
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the builtin, so that hipcc counts from here
    LV_TL(5)
#endif
    LV_TL_FLUSH
  }
}

// ---------------------------------------------------------------------------------------------
// One wave per SIMD: 4 waves x 64 query rows (two 32-row blocks per wave), the whole 512-entry
// register file per wave. Every K and V fragment read from LDS feeds two MFMAs (one per row block),
// half the LDS traffic per FLOP of the 32-rows-per-wave kernels. O^T and the Q^T fragments live in
// accumulator registers that only asm statements name (see mfma_s / mfma_o above); scores,
// probabilities and K/V fragments are ordinary hipcc values in the 256 vector registers.
//
// Software pipeline over K/V tiles, one workgroup barrier per tile. Iteration i runs two phases of
// 32 MFMAs, each MFMA followed by its share of the vector work (source order = issue order: a
// scheduling barrier closes every group, and an empty asm pins the arithmetic, which would
// otherwise sink to its first user):
//   QK phase   S(i+1) = K(i+1) Q^T          beside   P(i) second half = exp2(S(i) keys 32-63)
//   PV phase   O += V(i)^T P(i)             beside   row maxima of S(i+1), running-max decision,
//                                                    P(i+1) first half = exp2(S(i+1) keys 0-31)
// A rescale decided for tile i+1 during PV(i) scales l at once and O after PV(i) has been added
// (P(i) was formed at the old scale). K and V tiles are staged through registers into 2-slot LDS
// rings, each tile loaded one iteration before it is written and written one iteration before it
// is read: iteration i writes V(i+1) and requests V(i+2) in its QK phase, writes K(i+2) and
// requests K(i+3) in its PV phase. (LDS-DMA, which needs no registers, measured ~100 issue cycles
// per 1 KiB piece inside these phases, 8 pieces per wave and tile.)
template <bool PAGED>
__global__ __launch_bounds__(256, 1)
void prefill_w64_kernel(const PrefillParams p) {
  constexpr int NW = 4;
  constexpr int BM = 64 * NW;
  // Accumulator registers owned by the asm statements: O^T of row block rb at a[kAccO + 64 rb ..],
  // Q^T fragments of row block rb at a[kAccQ + 32 rb ..]. a[0:63] are left to hipcc, which places
  // the staging registers there (global loads and LDS stores take accumulator registers directly).
  constexpr int kAccO = 64, kAccQ = 192;
  // K slot s at s * kTileBytes, V slot s at (2 + s) * kTileBytes
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kTileBytes];
  lds_byte* const lds3 = (lds_byte*)lds;

  TileWork tw;
  if (!tile_of_block<BM>(p, tw)) return;   // launched with nsplit == 1
  const int kvh = tw.kvh, hq = tw.hq, q0 = tw.q0, b = tw.b;
  const int q_start = p.cu_q[b];
  const int Lq = p.cu_q[b + 1] - q_start;
  const int k_start = p.cu_k[b];
  const int Lk = p.cu_k[b + 1] - k_start;
  if (q0 >= Lq) return;
  const int coff = Lk - Lq;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5;
  const int c = lane & 31;
  const int wrow0 = q0 + wave * 64;
  const int my_lim0 = wrow0 + c + coff;        // last visible key of this lane's row in block 0
  const int my_lim1 = my_lim0 + 32;

  // ---- Q^T fragments -> a[128:191]; O^T a[0:127] = 0
  asm volatile("" ::: LV_ACC_CLOBBERS_64_127, LV_ACC_CLOBBERS_128_191, LV_ACC_CLOBBERS_192_255);
  static_for<0, 2>([&](auto rb_c) {
    constexpr int RBI = decltype(rb_c)::value;
    const int row_ld = min(wrow0 + 32 * RBI + c, Lq - 1);
    const uint16_t* qp = p.q + static_cast<int64_t>(q_start + row_ld) * p.q_stride0 + hq * kHeadDim + 8 * h;
    u32x4 qv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qv[s] = *reinterpret_cast<const u32x4*>(qp + 16 * s);
    static_for<0, 8>([&](auto s_c) {
      constexpr int S = decltype(s_c)::value;
      const u32 w0 = qv[S][0], w1 = qv[S][1], w2 = qv[S][2], w3 = qv[S][3];
      acc_write<kAccQ + 32 * RBI + 4 * S + 0>(__builtin_bit_cast(float, w0));
      acc_write<kAccQ + 32 * RBI + 4 * S + 1>(__builtin_bit_cast(float, w1));
      acc_write<kAccQ + 32 * RBI + 4 * S + 2>(__builtin_bit_cast(float, w2));
      acc_write<kAccQ + 32 * RBI + 4 * S + 3>(__builtin_bit_cast(float, w3));
    });
  });
  static_for<0, 128>([&](auto r_c) { acc_write<kAccO + decltype(r_c)::value>(0.f); });
  asm volatile("s_nop 3");

  const int last_row = min(q0 + BM - 1, Lq - 1);
  const int max_key = min(last_row + coff, Lk - 1);
  const int ntiles = max_key >= 0 ? max_key / BN + 1 : 0;      // tiles the workgroup walks
  // tiles this wave takes part in: those that start at or below its last row's limit
  int nact = 0;
  if (wrow0 < Lq && wrow0 + 63 + coff >= 0) nact = min(ntiles, ((wrow0 + 63 + coff) >> 6) + 1);

  // ---- LDS-DMA staging: wave w moves pieces w, w+4, w+8, w+12 (1 KiB = 4 tile rows each)
  const int32_t* bt_row = PAGED ? p.block_table + static_cast<int64_t>(b) * p.bt_stride : nullptr;
  const uint32_t k_row_bytes = static_cast<uint32_t>(p.k_stride0) * 2u;
  const uint32_t v_row_bytes = static_cast<uint32_t>(p.v_stride0) * 2u;
  const int lrow = lane >> 4;
  const int lchunk = (lane & 15) ^ ((lrow << 2) | (wave & 3));   // swizzle on the source side
  const uint32_t colb = static_cast<uint32_t>(kvh * kHeadDim + 8 * lchunk) * 2u;
  // per-lane byte offset of this lane's 16 bytes inside a tile, for each of the wave's 4 pieces
  uint32_t koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    koff[i] = static_cast<uint32_t>(4 * (wave + NW * i) + lrow) * k_row_bytes + colb;
    voff[i] = static_cast<uint32_t>(4 * (wave + NW * i) + lrow) * v_row_bytes + colb;
  }
  // global address of the first row of a tile (wave-uniform)
  typedef const __attribute__((address_space(1))) char* gptr;
  auto tile_base = [&](int tile, const uint16_t* src, uint32_t row_bytes) -> gptr {
    const int kt0 = tile * BN;
    int64_t base_row;
    if constexpr (PAGED) {
      const int blk = bt_row[kt0 / p.block_size];
      base_row = static_cast<int64_t>(blk) * p.block_size + (kt0 % p.block_size);
    } else {
      base_row = static_cast<int64_t>(k_start) + kt0;
    }
    // wave-uniform by construction; said so to hipcc, which must hold it in scalar registers
    const uint64_t a = reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(src) + base_row * row_bytes);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    // a pointer made from an integer is a flat pointer to hipcc; flat loads count in lgkmcnt as
    // well and would tie every LDS wait to the staging loads, so name the global address space
    return reinterpret_cast<gptr>((static_cast<uint64_t>(hi) << 32) | lo);
  };
  // piece I of a tile: 1 KiB, global (uniform base + per-lane offset) -> LDS (lane-linear at dst).
  // Register staging, one set of four 16-byte registers for K and one for V: a tile's four pieces
  // are loaded (global -> registers, uniform base + per-lane offset) a whole iteration before they
  // are written to LDS (lane-linear inside each 1 KiB piece; the swizzle is in the source offset):
  // the loads measured ~4000 cycles from issue to data under this kernel's own traffic.
  u32x4 kst[4], vst[4];
  auto stage_load = [&](u32x4 (&st)[4], gptr tb, const uint32_t (&off)[4], auto i_c) {
    constexpr int I = decltype(i_c)::value;
    if (LV_ABL & 8) return;
    st[I] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(tb + off[I]);
  };
  auto stage_write = [&](const u32x4 (&st)[4], int slot_off, auto i_c) {
    constexpr int I = decltype(i_c)::value;
    if (LV_ABL & 16) return;
    *reinterpret_cast<u32x4*>(lds + slot_off + 1024 * NW * I + 1024 * wave + 16 * lane) = st[I];
  };
  // The tile that runs past the sequence end is the last one the workgroup loads: from then on
  // the offsets point rows past the end at the sequence's last row (masked later).
  auto clamp_offsets = [&](int tile) {
    const int last = Lk - 1 - tile * BN;
    if (last < BN - 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t r = static_cast<uint32_t>(min(4 * (wave + NW * i) + lrow, last));
        koff[i] = r * k_row_bytes + colb;
        voff[i] = r * v_row_bytes + colb;
      }
    }
  };
  auto stage_tile_now = [&](int tile, const uint16_t* src, uint32_t row_bytes, const uint32_t (&off)[4], int slot_off) {
    gptr tb = tile_base(tile, src, row_bytes);
    static_for<0, 4>([&](auto i_c) { stage_load(kst, tb, off, i_c); });
    static_for<0, 4>([&](auto i_c) { stage_write(kst, slot_off, i_c); });
  };


  const float cexp = p.scale_log2e;
  const float defer_raw = kDeferLog2 / cexp;

  // ---- LDS read addresses inside slot 0 of K / of V (see prefill_attn_kernel)
  const int xr = ((c & 3) << 2) | ((c >> 2) & 3);
  const int gi = lane >> 4;
  const int tq = (lane & 15) >> 2;
  const int tp = lane & 3;
  lds_byte* kaddr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) kaddr[s] = lds3 + 256 * c + 16 * ((2 * s + h) ^ xr);
  lds_byte* vaddr[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int second = 0; second < 2; ++second) {
      const int row = 8 * second + 4 * (gi >> 1) + tq;
      const int ch = 4 * dt + 2 * (gi & 1) + (tp >> 1);
      vaddr[dt][second] = lds3 + lds_off(row, ch) + 8 * (tp & 1);   // + V slot offset at the read
    }
  // K fragments of d-steps S0 .. S0+3 of a 32-key half
  auto load_k4 = [&](bf16x8 (&kf)[4], int byte_off, auto s0_c) {
    constexpr int S0 = decltype(s0_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (LV_ABL & 32) { asm volatile("" : "=v"(kf[j])); continue; }
      kf[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kaddr[S0 + j] + byte_off);
    }
  };
  // V^T fragments of one 16-key step (four 32-column blocks)
  auto load_v4 = [&](bf16x8 (&vf)[4], int byte_off) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int second = 0; second < 2; ++second) {
        if (LV_ABL & 64) { if (second == 0) asm volatile("" : "=v"(vf[dt])); continue; }
        const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(vaddr[dt][second] + byte_off));
        vf[dt][4 * second + 0] = t[0];
        vf[dt][4 * second + 1] = t[1];
        vf[dt][4 * second + 2] = t[2];
        vf[dt][4 * second + 3] = t[3];
      }
  };

  LV_STAMP_DECL
  // ---- softmax state
  f32x16 S[2][2][2];        // [tile parity][32-key half][row block]
  u32 P[2][2][2][8];        // [tile parity][32-key half][row block] two bf16 per dword
  float m_run[2] = {-INFINITY, -INFINITY};
  float l_run[2] = {0.f, 0.f};
  float mc[2] = {0.f, 0.f};             // running max * scale * log2(e) the exponentials are taken against
  float pend_alpha[2] = {1.f, 1.f};
  bool pend[2] = {false, false};
  float ps[2] = {0.f, 0.f};             // row-sum shares collected since the last fold into l_run

  // The 64 probabilities a lane owes per tile (element e: 32-key half e >> 5, row block
  // (e >> 4) & 1, score register e & 15) go through a three-stage pipeline, one stage per MFMA gap,
  // so that no instruction in a gap waits for another one of the same gap:
  //   gap g    t  = S[e] * c - m * c
  //   gap g+1  pe = exp2(t)
  //   gap g+2  row sum += pe; an odd element is packed with its even partner into one dword
  float st_t = 0.f, st_pe = 0.f, st_even = 0.f;   // stage registers
  auto stream_step = [&](f32x16 (&sc)[2][2], u32 (&pw)[2][2][8], auto e_c) {
    constexpr int E = decltype(e_c)::value;       // 0..65 (64, 65 drain the pipeline)
    // entry pin: the stage registers pass through an asm statement that follows the gap's MFMA
    // in statement order, so hipcc cannot hoist this gap's vector work above that MFMA (it would
    // otherwise pair the MFMAs up and leave the second one waiting for the matrix pipe)
    asm volatile("" : "+v"(st_t), "+v"(st_pe), "+v"(st_even));
    if constexpr (E >= 2) {                       // third stage of element E - 2
      constexpr int E2 = E - 2, RB2 = (E2 >> 4) & 1, R2 = E2 & 15;
      ps[RB2] += st_pe;
      if constexpr ((R2 & 1) == 0) {
        st_even = st_pe;
      } else {
        bf16x2_t pk;
        pk[0] = static_cast<__bf16>(st_even);
        pk[1] = static_cast<__bf16>(st_pe);
        pw[E2 >> 5][RB2][R2 >> 1] = __builtin_bit_cast(u32, pk);
        asm volatile("" : "+v"(pw[E2 >> 5][RB2][R2 >> 1]));
      }
      asm volatile("" : "+v"(ps[RB2]), "+v"(st_even));
    }
    if constexpr (E >= 1 && E <= 64) {            // second stage of element E - 1
      st_pe = __builtin_amdgcn_exp2f(st_t);
      asm volatile("" : "+v"(st_pe));
    }
    if constexpr (E < 64) {                       // first stage of element E
      constexpr int RB0 = (E >> 4) & 1;
      st_t = fmaf(sc[E >> 5][RB0][E & 15], cexp, -mc[RB0]);
      asm volatile("" : "+v"(st_t));
    }
  };
  auto p_frag = [](const u32 (&pw)[8], int ks) -> bf16x8 {
    u32x4 t;
    t[0] = pw[4 * ks + 0]; t[1] = pw[4 * ks + 1]; t[2] = pw[4 * ks + 2]; t[3] = pw[4 * ks + 3];
    return __builtin_bit_cast(bf16x8, t);
  };
  auto max16 = [](const f32x16& s) -> float {
    float a[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a[j]) : "v"(s[3 * j]), "v"(s[3 * j + 1]), "v"(s[3 * j + 2]));
    float b0, b1;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b0) : "v"(a[0]), "v"(a[1]), "v"(a[2]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b1) : "v"(a[3]), "v"(a[4]), "v"(s[15]));
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(b0), "v"(b1));
    return m;
  };
  // causal mask of one tile's scores (diagonal tiles only; wave-uniform test)
  auto mask_tile = [&](int kt0, f32x16 (&sc)[2][2]) {
    if (kt0 + 63 > wrow0 + coff) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (key > my_lim0) sc[0][0][r] = -INFINITY;
        if (key + 32 > my_lim0) sc[1][0][r] = -INFINITY;
        if (key > my_lim1) sc[0][1][r] = -INFINITY;
        if (key + 32 > my_lim1) sc[1][1][r] = -INFINITY;
      }
    }
  };
  // row maxima of a tile for both row blocks (independent chains, interleaved), then the
  // running-max decisions (deferred rescale) under one wave-uniform test
  auto fmax_raw = [](float x, float y) -> float {     // v_max_f32 without canonicalising moves
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    return m;
  };
  auto decide_both = [&](f32x16 (&sc)[2][2]) {
    float mx[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) mx[rb] = fmax_raw(max16(sc[0][rb]), max16(sc[1][rb]));
    unsigned sw0[2], sw1[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const unsigned xb = __builtin_bit_cast(unsigned, mx[rb]);
      const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
      sw0[rb] = sw[0];
      sw1[rb] = sw[1];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
      mx[rb] = fmax_raw(__builtin_bit_cast(float, sw0[rb]), __builtin_bit_cast(float, sw1[rb]));
    const bool need0 = mx[0] - m_run[0] > defer_raw, need1 = mx[1] - m_run[1] > defer_raw;
    if (__builtin_amdgcn_ballot_w64(need0 || need1) != 0) {
      static_for<0, 2>([&](auto rb_c) {
        constexpr int RBI = decltype(rb_c)::value;
        if (__builtin_amdgcn_ballot_w64(RBI == 0 ? need0 : need1) != 0) {
          const float m_new = fmaxf(m_run[RBI], mx[RBI]);
          const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
          const float alpha = __builtin_amdgcn_exp2f((m_run[RBI] - m_use) * cexp);
          m_run[RBI] = m_new;
          l_run[RBI] = (l_run[RBI] + ps[RBI]) * alpha;   // everything summed so far is at the old scale
          ps[RBI] = 0.f;
          pend_alpha[RBI] = alpha;
          pend[RBI] = true;
          mc[RBI] = m_use * cexp;
        }
      });
    }
  };
  auto apply_pending = [&]() {
    static_for<0, 2>([&](auto rb_c) {
      constexpr int RBI = decltype(rb_c)::value;
      if (pend[RBI]) {        // wave-uniform (set under a ballot)
        asm volatile("s_nop 15");
        static_for<0, 64>([&](auto r_c) { acc_scale<kAccO + 64 * RBI + decltype(r_c)::value>(pend_alpha[RBI]); });
        asm volatile("s_nop 3");
        pend[RBI] = false;
      }
    });
  };
  auto settle = [&](f32x16 (&sc)[2][2]) {
    asm volatile("s_nop 15" : "+v"(sc[0][0]), "+v"(sc[0][1]), "+v"(sc[1][0]), "+v"(sc[1][1]));
  };

  // QK phase: scores of the tile in K slot KS into sc; vector work for gap g from fill(IntC<g>)
  auto qk_phase = [&](f32x16 (&sc)[2][2], auto kslot_c, auto&& pre, auto&& fill) {
    constexpr int KOFF = decltype(kslot_c)::value * kTileBytes;
    bf16x8 ka[4], kb[4];
    load_k4(ka, KOFF, IntC<0>{});
    load_k4(kb, KOFF, IntC<4>{});
    // the K tile became visible at the barrier just passed, so these reads cannot be issued any
    // earlier: vector work that does not need them runs under their latency
    pre();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 2>([&](auto half_c) {
      constexpr int HALF = decltype(half_c)::value;
      static_for<0, 8>([&](auto s_c) {
        constexpr int SI = decltype(s_c)::value;
        constexpr int G0 = 16 * HALF + 2 * SI;
        const bf16x8& kf = SI < 4 ? ka[SI & 3] : kb[SI & 3];
        mfma_s<kAccQ + 4 * SI, SI == 0>(sc[HALF][0], kf);
        fill(IntC<G0>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_s<kAccQ + 32 + 4 * SI, SI == 0>(sc[HALF][1], kf);
        fill(IntC<G0 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HALF == 0 && SI == 3) load_k4(ka, KOFF + 8192, IntC<0>{});
        if constexpr (HALF == 0 && SI == 7) load_k4(kb, KOFF + 8192, IntC<4>{});
      });
    });
  };
  // PV phase: O += V^T P for the tile in V slot VS with probabilities pr; fill as above
  auto pv_phase = [&](const u32 (&pr)[2][2][8], auto vslot_c, bf16x8 (&va)[4], bf16x8 (&vb)[4], auto&& fill) {
    constexpr int VOFF = decltype(vslot_c)::value * kTileBytes;
    asm volatile("s_nop 1");     // P was written by vector instructions
    static_for<0, 4>([&](auto ks_c) {
      constexpr int KS = decltype(ks_c)::value;     // 16-key step: half KS >> 1, step KS & 1
      bf16x8 pf0 = p_frag(pr[KS >> 1][0], KS & 1);
      bf16x8 pf1 = p_frag(pr[KS >> 1][1], KS & 1);
      // the fragments may have been gathered by vector moves just now: wait states before an
      // MFMA reads them
      asm volatile("s_nop 1" : "+v"(pf0), "+v"(pf1));
      static_for<0, 4>([&](auto dt_c) {
        constexpr int DT = decltype(dt_c)::value;
        constexpr int G0 = 8 * KS + 2 * DT;
        const bf16x8& vf = (KS & 1) ? vb[DT] : va[DT];
        mfma_o<kAccO + 16 * DT>(vf, pf0);
        fill(IntC<G0>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_o<kAccO + 64 + 16 * DT>(vf, pf1);
        fill(IntC<G0 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (KS == 0) load_v4(va, VOFF + 8192);
      if constexpr (KS == 1) load_v4(vb, VOFF + 12288);
    });
  };

  // Vector work of the PV phase (or of the prologue, stand-alone) for the tile whose scores were
  // just formed: gap 0 settles and masks them, gap 1 takes the row maxima and the running-max
  // decisions, gaps 3..30 the first stage of elements 0..27.
#ifndef LV_W64_SPLIT
#define LV_W64_SPLIT 28
#endif
  constexpr int kSplit = LV_W64_SPLIT;   // elements started in the PV phase (<= 28); the rest start in the QK phase
  auto start_fill = [&](int kt0, f32x16 (&sc)[2][2], u32 (&pw)[2][2][8], auto g_c) {
    constexpr int GP = decltype(g_c)::value;
    if constexpr (GP == 0) { settle(sc); mask_tile(kt0, sc); }
    if constexpr (GP == 1) decide_both(sc);
    if constexpr (GP >= 3 && GP < 3 + kSplit) stream_step(sc, pw, IntC<GP - 3>{});
  };
  // Vector work of the QK phase: elements kSplit..63 of the previous tile's scores: the first
  // kLeft - 32 under the latency of the phase's first K reads (finish_pre), then one per gap, then
  // the two steps that drain the pipeline.
  constexpr int kPre = 64 - kSplit - 32;
  auto finish_pre = [&](f32x16 (&sc)[2][2], u32 (&pw)[2][2][8]) {
    static_for<0, kPre>([&](auto j_c) { stream_step(sc, pw, IntC<kSplit + decltype(j_c)::value>{}); });
  };
  auto finish_fill = [&](f32x16 (&sc)[2][2], u32 (&pw)[2][2][8], auto g_c) {
    constexpr int GP = decltype(g_c)::value;
    stream_step(sc, pw, IntC<kSplit + kPre + GP>{});
    if constexpr (GP == 31) {
      stream_step(sc, pw, IntC<64>{});
      stream_step(sc, pw, IntC<65>{});
    }
  };

  // iteration i (parity PAR): QK(i+1) beside the second half of softmax(i), then PV(i) beside the
  // first half of softmax(i+1)
  auto iteration = [&](int i, auto par_c) {
    constexpr int PAR = decltype(par_c)::value, NPAR = PAR ^ 1;
    // On entry the V set holds V(i+1) and the K set K(i+2) (requested an iteration ago).
    // QK phase: V(i+1) -> V slot NPAR (V(i-1) was read an iteration ago), V(i+2) requested.
    // PV phase: K(i+2) -> K slot PAR (K(i) was read an iteration ago), K(i+3) requested.
    // Past the last tile the last tile is requested and written again (into slots nobody reads
    // any more): the memory operations are then the same in every iteration, with no branch
    // around them, and hipcc can count its vmcnt waits exactly (a wait for the set loaded an
    // iteration ago, not for the one requested half an iteration ago).
    gptr vtb = tile_base((LV_ABL & 8192) ? 0 : min(i + 2, ntiles - 1), p.v, v_row_bytes);
    gptr ktb = tile_base((LV_ABL & 8192) ? 0 : min(i + 3, ntiles - 1), p.k, k_row_bytes);
    if (i < nact) {
      LV_STAMP_COUNT
      bf16x8 va[4], vb[4];      // V^T fragments of the first two 16-key steps of V(i)
      qk_phase(S[NPAR], IntC<NPAR>{}, [&]() { finish_pre(S[PAR], P[PAR]); }, [&](auto g_c) {
        constexpr int GP = decltype(g_c)::value;
        // staging traffic sits in the middle of the phase, away from the latency-critical reads
        // at its start (LDS stores are slow, ~13 cycles of the store path each, and reads queue
        // behind them)
        if constexpr (GP >= 10 && GP < 14) stage_write(vst, (2 + NPAR) * kTileBytes, IntC<GP - 10>{});
        if constexpr (GP >= 14 && GP < 18) stage_load(vst, vtb, voff, IntC<GP - 14>{});
        if constexpr (GP == 26) load_v4(va, (2 + PAR) * kTileBytes);
        finish_fill(S[PAR], P[PAR], g_c);
      });
      __builtin_amdgcn_sched_barrier(0);
      load_v4(vb, (2 + PAR) * kTileBytes + 4096);
      pv_phase(P[PAR], IntC<2 + PAR>{}, va, vb, [&](auto g_c) {
        constexpr int GP = decltype(g_c)::value;
        if constexpr (GP >= 18 && GP < 22) stage_write(kst, PAR * kTileBytes, IntC<GP - 18>{});
        if constexpr (GP == 22) clamp_offsets(min(i + 3, ntiles - 1));
        if constexpr (GP >= 22 && GP < 26) stage_load(kst, ktb, koff, IntC<GP - 22>{});
        start_fill((i + 1) * BN, S[NPAR], P[NPAR], g_c);
      });
      __builtin_amdgcn_sched_barrier(0);
      apply_pending();
    } else {
      static_for<0, 4>([&](auto i_c) { stage_write(vst, (2 + NPAR) * kTileBytes, i_c); });
      static_for<0, 4>([&](auto i_c) { stage_load(vst, vtb, voff, i_c); });
      static_for<0, 4>([&](auto i_c) { stage_write(kst, PAR * kTileBytes, i_c); });
      clamp_offsets(min(i + 3, ntiles - 1));
      static_for<0, 4>([&](auto i_c) { stage_load(kst, ktb, koff, i_c); });
    }
    LV_STAMP(1)
    if (!(LV_ABL & 256)) __syncthreads();
    LV_STAMP(0)
  };

  if (ntiles > 0) {
    clamp_offsets(0);
    stage_tile_now(0, p.k, k_row_bytes, koff, 0);
    stage_tile_now(0, p.v, v_row_bytes, voff, 2 * kTileBytes);
    {
      // K(1) into its slot; V(1) and K(2) stay in the staging sets (tile indices clamped as in
      // the iterations)
      const int t1 = min(1, ntiles - 1), t2 = min(2, ntiles - 1);
      clamp_offsets(t1);
      stage_tile_now(t1, p.k, k_row_bytes, koff, kTileBytes);
      gptr vtb1 = tile_base(t1, p.v, v_row_bytes);
      static_for<0, 4>([&](auto i_c) { stage_load(vst, vtb1, voff, i_c); });
      clamp_offsets(t2);
      gptr ktb2 = tile_base(t2, p.k, k_row_bytes);
      static_for<0, 4>([&](auto i_c) { stage_load(kst, ktb2, koff, i_c); });
    }
    __syncthreads();
    if (nact > 0) {
      qk_phase(S[0], IntC<0>{}, [&]() {}, [&](auto) {});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 32>([&](auto g_c) {
        start_fill(0, S[0], P[0], g_c);
        __builtin_amdgcn_sched_barrier(0);
      });
      apply_pending();     // first decision: alpha = 0 on a zero accumulator; clears the flag
    }
    __syncthreads();       // every wave has read K(0) before iteration 0 overwrites its slot
    LV_STAMP(14)
    for (int i = 0; i < ntiles; i += 2) {
      iteration(i, IntC<0>{});
      if (i + 1 < ntiles) iteration(i + 1, IntC<1>{});
    }
  }

  LV_STAMP_FLUSH
  // ---- epilogue: normalise and store (d = 32*dt + (r&3) + 8*(r>>2) + 4*h)
  asm volatile("s_nop 15");
  static_for<0, 2>([&](auto rb_c) {
    constexpr int RBI = decltype(rb_c)::value;
    const float l_lane = l_run[RBI] + ps[RBI];
    const float l_tot = l_lane + __shfl_xor(l_lane, 32);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int my_row = wrow0 + 32 * RBI + c;
    uint16_t* ob = p.o + static_cast<int64_t>(q_start + min(my_row, Lq - 1)) * p.o_stride0 + hq * kHeadDim;
    u32x2 w[16];
    static_for<0, 4>([&](auto dt_c) {
      static_for<0, 4>([&](auto rg_c) {
        constexpr int DT = decltype(dt_c)::value, RG = decltype(rg_c)::value;
        constexpr int R0 = kAccO + 64 * RBI + 16 * DT + 4 * RG;
        w[4 * DT + RG][0] = pack_bf16x2(acc_read<R0 + 0>() * inv, acc_read<R0 + 1>() * inv);
        w[4 * DT + RG][1] = pack_bf16x2(acc_read<R0 + 2>() * inv, acc_read<R0 + 3>() * inv);
      });
    });
    store_row_128(ob, h, my_row < Lq, w);
  });
}

// Combines the nsplit runs of every (query row, head): out = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s.
// 32 lanes per (row, head), four columns each; eight pairs per workgroup.
__global__ __launch_bounds__(256) void prefill_merge_kernel(const float* __restrict__ ws_o,
                                                            const float* __restrict__ ws_ml,
                                                            uint16_t* __restrict__ o, int64_t o_stride0,
                                                            int num_heads, int nsplit, int64_t pairs,
                                                            const int32_t* __restrict__ cu_q, int batch) {
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5);
  if (pair >= pairs) return;
  const int ln = threadIdx.x & 31;
  const int64_t row = pair / num_heads;
  // rows of q past the last sequence (a padded q) have no partials: like the one-pass kernels,
  // leave their output rows untouched
  if (row >= cu_q[batch]) return;
  const int hq = static_cast<int>(pair % num_heads);
  const int64_t base = pair * nsplit;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ws_ml[(base + s) * 2]);
  const float Mu = (M == -INFINITY) ? 0.f : M;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float L = 0.f;
#pragma unroll 4
  for (int s = 0; s < nsplit; ++s) {
    const float2 ml = *reinterpret_cast<const float2*>(ws_ml + (base + s) * 2);
    const float4 v = *reinterpret_cast<const float4*>(ws_o + (base + s) * kHeadDim + 4 * ln);
    const float sc = __builtin_amdgcn_exp2f(ml.x - Mu);
    acc.x += sc * v.x;
    acc.y += sc * v.y;
    acc.z += sc * v.z;
    acc.w += sc * v.w;
    L += sc * ml.y;
  }
  const float inv = L > 0.f ? 1.f / L : 0.f;
  u32x2 w;
  w[0] = pack_bf16x2(acc.x * inv, acc.y * inv);
  w[1] = pack_bf16x2(acc.z * inv, acc.w * inv);
  *reinterpret_cast<u32x2*>(o + row * o_stride0 + hq * kHeadDim + 4 * ln) = w;
}

// (sequence, query tile) items of the launch: the caller's table (empty entries included), or every
// tile of every sequence up to max_seqlen_q
static int64_t grid_items(const PrefillParams& p, int bm) {
  if (p.tile_order) return p.order_len;
  return static_cast<int64_t>((p.max_seqlen_q + bm - 1) / bm) * p.batch;
}

template <bool PAGED>
static void launch_w64(const PrefillParams& p, hipStream_t stream) {
  const int64_t total = grid_items(p, 256) * p.num_heads;
  hipLaunchKernelGGL((prefill_w64_kernel<PAGED>), dim3(static_cast<unsigned>(total)), dim3(256), 0, stream, p);
}

template <bool PAGED>
static void launch_pp(const PrefillParams& p, hipStream_t stream) {
  if (p.order_lanes > 0) {
    // item table: one workgroup per (query head, table column, lane), each walking its lane's entries
    const unsigned grid = static_cast<unsigned>(p.num_heads) * p.order_cols * p.order_lanes;
    hipLaunchKernelGGL((prefill_pp_kernel<PAGED, true>), dim3(grid), dim3(512), 0, stream, p);
    return;
  }
  const int64_t total = grid_items(p, 256) * p.num_heads;
  hipLaunchKernelGGL((prefill_pp_kernel<PAGED, false>), dim3(static_cast<unsigned>(total)), dim3(512), 0, stream, p);
}

template <bool PAGED, int NW, int RB, bool DMA>
static void launch_cfg(const PrefillParams& p, hipStream_t stream) {
  constexpr int BM = 32 * RB * NW;
  const int64_t total = grid_items(p, BM) * p.num_heads * p.nsplit;
  hipLaunchKernelGGL((prefill_attn_kernel<PAGED, NW, RB, DMA>), dim3(static_cast<unsigned>(total)), dim3(64 * NW), 0,
                     stream, p);
}

// Kernel configuration of a launch: 0 = W4 (128-row tiles), 1 = W8, 2 = PP, 4 = W64 (256-row tiles);
// -1 for an unknown hint. KV splits live in the free-running kernels only: the 8-wave one stands in
// for the two long-prompt kernels (within 3-7 % of them unsplit).
static int prefill_cfg(int max_seqlen_q, int nsplit, int kernel_hint) {
  int cfg = max_seqlen_q >= 2048 ? 2 : (max_seqlen_q > 1024 ? 1 : 0);
  switch (kernel_hint & 0xff) {
    case LV_PREFILL_KERNEL_AUTO: break;
    case LV_PREFILL_KERNEL_W4: cfg = 0; break;
    case LV_PREFILL_KERNEL_W8: cfg = 1; break;
    case LV_PREFILL_KERNEL_PP: cfg = 2; break;
    case LV_PREFILL_KERNEL_W64: cfg = 4; break;
    default: return -1;
  }
  if (nsplit > 1 && (cfg == 2 || cfg == 4)) cfg = 1;
  return cfg;
}

int prefill_tile_rows(int max_seqlen_q, int num_splits, int kernel_hint) {
  const int cfg = prefill_cfg(max_seqlen_q, num_splits, kernel_hint);
  return cfg < 0 ? 0 : (cfg == 0 ? 128 : 256);
}

int prefill_order_columns(int num_kv_heads) {
  return (num_kv_heads == 1 || num_kv_heads == 2 || num_kv_heads == 4) ? 8 / num_kv_heads : 1;
}

int prefill_order_lanes(int num_heads, int num_kv_heads) {
  const int per_lane = num_heads * prefill_order_columns(num_kv_heads);   // workgroups one lane adds to the grid
  return per_lane >= 256 ? 1 : 256 / per_lane;                            // 256 compute units, one 512-thread workgroup each
}

int launch_prefill_attn(const PrefillParams& p, hipStream_t stream) {
  if (grid_items(p, 128) * p.num_heads * p.nsplit > INT32_MAX)
    return fail(LV_ERR_UNSUPPORTED, "prefill: grid too large");
  // cfg 0: 4 waves x 32 rows = 128-row query tiles, two workgroups per CU (short batches keep the
  //        grid wide);  cfg 1: 8 waves x 32 rows = 256-row tiles, half the K/V staging per row;
  // cfg 2: the phase-locked 8-wave kernel (long prompts);
  // cfg 4: 4 waves x 64 rows, one wave per SIMD, asm-owned accumulator registers (opt-in).
  // The caller's kernel_hint (LV_PREFILL_KERNEL_*, optionally | LV_PREFILL_STAGE_DMA) overrides
  // the choice by prompt length.
  const int dma = (p.kernel_hint & LV_PREFILL_STAGE_DMA) != 0;
  const int cfg = prefill_cfg(p.max_seqlen_q, p.nsplit, p.kernel_hint);
  if (cfg < 0) return fail(LV_ERR_INVALID_ARGUMENT, "prefill: kernel_hint %d", p.kernel_hint);
  if (p.tile_order && p.order_rows != (cfg == 0 ? 128 : 256))
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: tile_order built for %d-row tiles, this launch uses %d-row tiles "
                "(lv_prefill_tile_rows)", p.order_rows, cfg == 0 ? 128 : 256);
  if (p.order_lanes > 0 && cfg != 2)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: an item table (tile_order_lanes > 0) is for the phase-locked kernel "
                "(max_seqlen_q >= 2048 or LV_PREFILL_KERNEL_PP), one KV pass");
  if (cfg == 4) {
    if (p.block_table) launch_w64<true>(p, stream); else launch_w64<false>(p, stream);
  } else if (cfg == 2) {
    if (p.block_table) launch_pp<true>(p, stream); else launch_pp<false>(p, stream);
  } else if (p.block_table) {
    if (cfg == 1) { if (dma) launch_cfg<true, 8, 1, true>(p, stream); else launch_cfg<true, 8, 1, false>(p, stream); }
    else { if (dma) launch_cfg<true, 4, 1, true>(p, stream); else launch_cfg<true, 4, 1, false>(p, stream); }
  } else {
    if (cfg == 1) { if (dma) launch_cfg<false, 8, 1, true>(p, stream); else launch_cfg<false, 8, 1, false>(p, stream); }
    else { if (dma) launch_cfg<false, 4, 1, true>(p, stream); else launch_cfg<false, 4, 1, false>(p, stream); }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "prefill_attn launch: %s", hipGetErrorString(e));
  if (p.nsplit > 1) {
    const int64_t pairs = p.total_q * p.num_heads;
    hipLaunchKernelGGL(prefill_merge_kernel, dim3(static_cast<unsigned>((pairs + 7) / 8)), dim3(256), 0, stream,
                       p.ws_o, p.ws_ml, p.o, p.o_stride0, p.num_heads, p.nsplit, pairs, p.cu_q, p.batch);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "prefill_merge launch: %s", hipGetErrorString(e));
  }
  return LV_OK;
}

}  // namespace lv
