// valu_rate.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU/LDS instructions k_knn is made
// of, measured on the device: decides which instruction mixes are worth hand-writing (DESIGN.md "Roofline").
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Each kernel runs ITER iterations of 32 independent copies of one instruction; grid = CUs x 4 SIMDs x W waves.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

constexpr int ITER = 4096;

#define REP4(s) s s s s
#define REP8(s) REP4(s) REP4(s)

// 8 accumulators a0..a7 (32-bit) / d0..d7 (64-bit), 4 rounds per iteration = 32 instructions
#define KERNEL32(name, body)                                                                             \
    __global__ __launch_bounds__(64) void name(float* out, float seed)                                    \
    {                                                                                                    \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
              a6 = a0 + 6, a7 = a0 + 7, b = seed * 0.5f, c = seed * 0.25f;                                \
        for (int i = 0; i < ITER; ++i) {                                                                 \
            asm volatile(REP4(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), \
                         "+v"(a7)                                                                        \
                         : "v"(b), "v"(c)                                                                \
                         : "vcc");                                                                       \
        }                                                                                                \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
    }

#define KERNEL64(name, body)                                                                             \
    __global__ __launch_bounds__(64) void name(float* out, float seed)                                    \
    {                                                                                                    \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
               a6 = a0 + 6, a7 = a0 + 7, b = seed * 0.5, c = seed * 0.25;                                 \
        for (int i = 0; i < ITER; ++i) {                                                                 \
            asm volatile(REP4(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), \
                         "+v"(a7)                                                                        \
                         : "v"(b), "v"(c)                                                                \
                         : "vcc");                                                                       \
        }                                                                                                \
        out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);   \
    }

#define EACH8(op, tail)                                                                                          \
    op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n" op " %3, %3, " tail "\n" op " %4, %4, " tail \
       "\n" op " %5, %5, " tail "\n" op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"

KERNEL32(k_mul_f32, EACH8("v_mul_f32", "%8"))
KERNEL32(k_add_f32, EACH8("v_add_f32", "%8"))
KERNEL32(k_max3_f32, EACH8("v_max3_f32", "%8, %9"))
KERNEL32(k_med3_f32, EACH8("v_med3_f32", "%8, %9"))
KERNEL32(k_cndmask, EACH8("v_cndmask_b32", "%8, vcc"))
// v_cndmask variants: mask in an SGPR pair (VOP3), mask in VCC written once outside the loop, destinations
// different from the sources
__global__ __launch_bounds__(64) void k_cndmask_sgpr(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 0.5f;
    unsigned long long mask = 0x5555555555555555ull;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %4, %5\nv_cndmask_b32_e64 %1, %1, %4, %5\n"
                          "v_cndmask_b32_e64 %2, %2, %4, %5\nv_cndmask_b32_e64 %3, %3, %4, %5\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "v"(b), "s"(mask));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void k_cndmask_vcc_set(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 0.5f;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 vcc, 0x55555555\n" REP8("v_cndmask_b32_e32 %0, %0, %4, vcc\nv_cndmask_b32_e32 %1, %1, %4, vcc\n"
                                                        "v_cndmask_b32_e32 %2, %2, %4, vcc\nv_cndmask_b32_e32 %3, %3, %4, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "v"(b)
                     : "vcc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void k_cndmask_fresh_dst(float* out, float seed)
{
    float a0, a1, a2, a3, b = seed * 0.5f, c = seed + threadIdx.x;
    unsigned long long mask = 0x5555555555555555ull;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_cndmask_b32_e64 %0, %4, %5, %6\nv_cndmask_b32_e64 %1, %4, %5, %6\n"
                          "v_cndmask_b32_e64 %2, %4, %5, %6\nv_cndmask_b32_e64 %3, %4, %5, %6\n")
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                     : "v"(b), "v"(c), "s"(mask));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// bit-arithmetic alternative to "x = cond ? x : PAD": v_bfi_b32 and v_or_b32
KERNEL32(k_bfi, EACH8("v_bfi_b32", "%8, %9"))
KERNEL32(k_or, EACH8("v_or_b32", "%8"))
KERNEL32(k_ashr, EACH8("v_ashrrev_i32", "%8"))
KERNEL32(k_min_u32, EACH8("v_min_u32", "%8"))
KERNEL32(k_min3_u32, EACH8("v_min3_u32", "%8, %9"))
KERNEL32(k_add_u32, EACH8("v_add_u32", "%8"))
KERNEL32(k_lshl_add, EACH8("v_lshl_add_u32", "%8, %9"))

KERNEL32(k_fmac_f32, EACH8("v_fmac_f32_e32", "%8"))
KERNEL32(k_fma_f32, EACH8("v_fma_f32", "%8, %9"))
KERNEL32(k_max_f32, EACH8("v_max_f32_e32", "%8"))
KERNEL32(k_cmp_f32, REP8("v_cmp_le_f32 vcc, %0, %8\n"))
KERNEL32(k_mov_b32, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, "
                    "%8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n")
KERNEL64(k_pk_mul_f32, EACH8("v_pk_mul_f32", "%8"))
KERNEL64(k_pk_add_f32, EACH8("v_pk_add_f32", "%8"))
KERNEL64(k_pk_fma_f32, EACH8("v_pk_fma_f32", "%8, %9"))
KERNEL64(k_min_f64, EACH8("v_min_f64", "%8"))
KERNEL64(k_max_f64, EACH8("v_max_f64", "%8"))
KERNEL64(k_add_f64, EACH8("v_add_f64", "%8"))
KERNEL64(k_cmp_u64, REP8("v_cmp_lt_u64 vcc, %0, %8\n"))
KERNEL64(k_mov_b64, "v_mov_b64 %0, %8\nv_mov_b64 %1, %8\nv_mov_b64 %2, %8\nv_mov_b64 %3, %8\nv_mov_b64 %4, %8\nv_mov_b64 %5, "
                    "%8\nv_mov_b64 %6, %8\nv_mov_b64 %7, %8\n")
KERNEL64(k_pk_mov_b32, EACH8("v_pk_mov_b32", "%8"))

// scalar ALU: 32 s_add_u32 per iteration on 8 SGPRs
__global__ __launch_bounds__(64) void k_salu(float* out, float seed)
{
    int s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("s_add_u32 %0, %0, 3\ns_add_u32 %1, %1, 5\ns_add_u32 %2, %2, 7\ns_add_u32 %3, %3, 9\n")
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                     :
                     : "scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = seed + (s0 + s1 + s2 + s3);
}

// VALU and SALU interleaved 1:1 (does scalar issue overlap vector issue of the same wave / other waves?)
__global__ __launch_bounds__(64) void k_valu_salu(float* out, float seed)
{
    int s0 = blockIdx.x, s1 = s0 + 1;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, b = seed * 0.5f;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_mul_f32 %0, %0, %4\ns_add_u32 %2, %2, 3\nv_mul_f32 %1, %1, %4\ns_add_u32 %3, %3, 5\n")
                     : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1)
                     : "v"(b)
                     : "scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + (s0 + s1);
}

// LDS: ds_write2_b32 to the lane's own column (stride 64 dwords), 16 per iteration
__global__ __launch_bounds__(64) void k_ds_write2(float* out, float seed)
{
    __shared__ float buf[64 * 16];
    float a0 = seed + threadIdx.x;
    unsigned addr = threadIdx.x * 4;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("ds_write2_b32 %0, %1, %1 offset1:64\nds_write2_b32 %0, %1, %1 offset0:128 offset1:192\n")
                     :
                     : "v"(addr), "v"(a0)
                     : "memory");
    }
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x];
}

// ds_write2_b32 with EXEC narrowed to `lanes` lanes (0, 1 or 4): what does a mostly rejected candidate cost?
template <int LANES>
__global__ __launch_bounds__(64) void k_ds_write2_masked(float* out, float seed)
{
    __shared__ float buf[64 * 16];
    float a0 = seed + threadIdx.x;
    unsigned addr = threadIdx.x * 4;
    const unsigned long long mask = LANES == 0 ? 0ull : (LANES == 1 ? 1ull : 0x1111ull);
    unsigned long long saved;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\t" REP8(
                         "ds_write2_b32 %1, %2, %2 offset1:64\nds_write2_b32 %1, %2, %2 offset0:128 offset1:192\n") "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "v"(addr), "v"(a0), "s"(mask)
                     : "memory");
    }
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x];
}

// the accept sequence of k_knn: 2 x v_cmpx, ds_write2 under the narrowed EXEC, address bump, EXEC restore, pos bump
// (tau = -1: every lane rejects, the common case); 8 per iteration like one leaf
__global__ __launch_bounds__(64) void k_accept_seq(float* out, float seed)
{
    __shared__ float buf[64 * 16];
    float d2 = seed + threadIdx.x, tau = -1.f, m = 1.f;
    unsigned wa = threadIdx.x * 4, pos = 0;
    unsigned long long saved;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 %[sv], exec\n\t" REP8("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                                                     "v_cmpx_le_f32_e32 %[eps], %[m]\n\t"
                                                     "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                                                     "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                                                     "s_mov_b64 exec, %[sv]\n\t"
                                                     "v_add_u32_e32 %[pos], 1, %[pos]\n\t")
                     : [wa] "+v"(wa), [pos] "+v"(pos), [sv] "=&s"(saved)
                     : [d2] "v"(d2), [tau] "v"(tau), [m] "v"(m), [eps] "s"(seed)
                     : "vcc", "memory");
    }
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x] + wa + pos;
}
// the same without the LDS write (what do the EXEC round trips cost by themselves?)
__global__ __launch_bounds__(64) void k_accept_seq_nolds(float* out, float seed)
{
    float d2 = seed + threadIdx.x, tau = -1.f, m = 1.f;
    unsigned wa = threadIdx.x * 4, pos = 0;
    unsigned long long saved;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 %[sv], exec\n\t" REP8("v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                                                     "v_cmpx_le_f32_e32 %[eps], %[m]\n\t"
                                                     "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                                                     "s_mov_b64 exec, %[sv]\n\t"
                                                     "v_add_u32_e32 %[pos], 1, %[pos]\n\t")
                     : [wa] "+v"(wa), [pos] "+v"(pos), [sv] "=&s"(saved)
                     : [d2] "v"(d2), [tau] "v"(tau), [m] "v"(m), [eps] "s"(seed)
                     : "vcc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = wa + pos;
}
// one candidate of a leaf as k_knn issues it: 8 distance ops + max3 + the accept sequence (13 VALU + 1 LDS + 1 SALU)
__global__ __launch_bounds__(64) void k_candidate(float* out, float seed)
{
    __shared__ float buf[64 * 16];
    float qx = seed + threadIdx.x, qy = qx * 0.5f, qz = qx * 0.25f, tau = -1.f;
    unsigned wa = threadIdx.x * 4, pos = 0;
    unsigned long long saved;
    float dx, dy, dz, d2, t, m;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 %[sv], exec\n\t" REP8("v_sub_f32_e32 %[dx], %[px], %[qx]\n\t"
                                                     "v_sub_f32_e32 %[dy], %[px], %[qy]\n\t"
                                                     "v_sub_f32_e32 %[dz], %[px], %[qz]\n\t"
                                                     "v_mul_f32_e32 %[d2], %[dx], %[dx]\n\t"
                                                     "v_mul_f32_e32 %[t], %[dy], %[dy]\n\t"
                                                     "v_add_f32_e32 %[d2], %[d2], %[t]\n\t"
                                                     "v_mul_f32_e32 %[t], %[dz], %[dz]\n\t"
                                                     "v_add_f32_e32 %[d2], %[d2], %[t]\n\t"
                                                     "v_max3_f32 %[m], |%[dx]|, |%[dy]|, |%[dz]|\n\t"
                                                     "v_cmpx_le_f32_e32 %[d2], %[tau]\n\t"
                                                     "v_cmpx_le_f32_e32 %[px], %[m]\n\t"
                                                     "ds_write2_b32 %[wa], %[pos], %[d2] offset1:1\n\t"
                                                     "v_add_u32_e32 %[wa], 0x200, %[wa]\n\t"
                                                     "s_mov_b64 exec, %[sv]\n\t"
                                                     "v_add_u32_e32 %[pos], 1, %[pos]\n\t")
                     : [wa] "+v"(wa), [pos] "+v"(pos), [sv] "=&s"(saved), [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz),
                       [d2] "=&v"(d2), [t] "=&v"(t), [m] "=&v"(m)
                     : [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz), [tau] "v"(tau), [px] "s"(seed)
                     : "vcc", "memory");
    }
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x] + wa + pos;
}

// lane exchanges through the scalar file, as the point-per-lane leaf form of k_knn uses them (pcpx_query.hip: sparse_leaf)
__global__ __launch_bounds__(64) void k_readlane(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned s0, s1, s2, s3, acc = 0;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b32 m0, 5\n" REP8("v_readlane_b32 %0, %4, m0\nv_readlane_b32 %1, %5, m0\n"
                                             "v_readlane_b32 %2, %6, m0\nv_readlane_b32 %3, %7, m0\n")
                     : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                     : "m0");
        acc += s0 ^ s1 ^ s2 ^ s3;
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + acc;
}
// each read followed by a vector instruction that takes the scalar just written as an operand
__global__ __launch_bounds__(64) void k_readlane_use(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned s0, s1, s2, s3;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b32 m0, 5\n" REP4("v_readlane_b32 %0, %4, m0\nv_readlane_b32 %1, %5, m0\n"
                                             "v_readlane_b32 %2, %6, m0\nv_readlane_b32 %3, %7, m0\n"
                                             "v_subrev_f32_e32 %4, %0, %4\nv_subrev_f32_e32 %5, %1, %5\n"
                                             "v_subrev_f32_e32 %6, %2, %6\nv_subrev_f32_e32 %7, %3, %7\n")
                     : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     :
                     : "m0");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void k_writelane(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b32 m0, 5\n" REP8("v_writelane_b32 %0, 7, m0\nv_writelane_b32 %1, 7, m0\n"
                                             "v_writelane_b32 %2, 7, m0\nv_writelane_b32 %3, 7, m0\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     :
                     : "m0");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void k_mbcnt(float* out, float seed)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned s0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(seed) | 0xf0f0u);
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_mbcnt_lo_u32_b32 %0, %4, %0\nv_mbcnt_lo_u32_b32 %1, %4, %1\n"
                          "v_mbcnt_lo_u32_b32 %2, %4, %2\nv_mbcnt_lo_u32_b32 %3, %4, %3\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "s"(s0));
    }
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(a0 + a1 + a2 + a3);
}
// one needing lane of a sparse leaf as k_knn issues it: 5 v_readlane, 8 distance ops, v_cmpx, rank, address, LDS write,
// v_writelane (17 VALU + 1 LDS + 8 SALU + the loop branch, which is left out here)
__global__ __launch_bounds__(64) void k_sparse_owner(float* out, float seed)
{
    __shared__ float buf[64 * 16];
    float qx = seed + threadIdx.x, qy = qx * 0.5f, qz = qx * 0.25f, tau = -1.f, cx = seed, cy = seed * 2, cz = seed * 3;
    unsigned wa = threadIdx.x * 4, pos = threadIdx.x;
    unsigned long long saved, todo = ~0ull;
    unsigned ox, oy, oz, otau, owa, owner;
    float d, e;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("s_mov_b64 %[sv], exec\n\t" REP8("s_ff1_i32_b64 m0, %[todo]\n\t"
                                                     "s_bitset1_b64 %[todo], m0\n\t"
                                                     "v_readlane_b32 %[sx], %[qx], m0\n\t"
                                                     "v_readlane_b32 %[sy], %[qy], m0\n\t"
                                                     "v_readlane_b32 %[sz], %[qz], m0\n\t"
                                                     "v_readlane_b32 %[st], %[tau], m0\n\t"
                                                     "v_readlane_b32 %[sw], %[wa], m0\n\t"
                                                     "s_mov_b64 exec, 0xff\n\t"
                                                     "v_subrev_f32_e32 %[d], %[sx], %[cx]\n\t"
                                                     "v_subrev_f32_e32 %[e], %[sy], %[cy]\n\t"
                                                     "v_mul_f32_e32 %[d], %[d], %[d]\n\t"
                                                     "v_mul_f32_e32 %[e], %[e], %[e]\n\t"
                                                     "v_add_f32_e32 %[d], %[d], %[e]\n\t"
                                                     "v_subrev_f32_e32 %[e], %[sz], %[cz]\n\t"
                                                     "v_mul_f32_e32 %[e], %[e], %[e]\n\t"
                                                     "v_add_f32_e32 %[d], %[d], %[e]\n\t"
                                                     "v_cmpx_ge_f32_e32 %[st], %[d]\n\t"
                                                     "s_bcnt1_i32_b64 %[L], exec\n\t"
                                                     "v_mbcnt_lo_u32_b32 %[e], exec_lo, 0\n\t"
                                                     "v_lshl_add_u32 %[e], %[e], 9, %[sw]\n\t"
                                                     "ds_write2_b32 %[e], %[pos], %[d] offset1:1\n\t"
                                                     "s_lshl_b32 %[L], %[L], 9\n\t"
                                                     "s_add_u32 %[sw], %[sw], %[L]\n\t"
                                                     "s_cmp_lg_u64 %[todo], 0\n\t"
                                                     "v_writelane_b32 %[wa], %[sw], m0\n\t"
                                                     "s_mov_b64 exec, %[sv]\n\t")
                     : [sv] "=&s"(saved), [L] "=&s"(owner), [sx] "=&s"(ox), [sy] "=&s"(oy), [sz] "=&s"(oz), [st] "=&s"(otau),
                       [sw] "=&s"(owa), [d] "=&v"(d), [e] "=&v"(e), [wa] "+v"(wa), [todo] "+s"(todo)
                     : [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz), [tau] "v"(tau), [cx] "v"(cx), [cy] "v"(cy), [cz] "v"(cz), [pos] "v"(pos)
                     : "m0", "vcc", "scc", "memory");
    }
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x] + wa;
}

#define LDS_KERNEL(name, body, clob)                                                  \
    __global__ __launch_bounds__(64) void name(float* out, float seed)                 \
    {                                                                                 \
        __shared__ float buf[64 * 16];                                                \
        double a0 = seed + threadIdx.x;                                               \
        unsigned addr = threadIdx.x * 8;                                              \
        for (int i = 0; i < ITER; ++i) {                                              \
            asm volatile(REP8(body) : "+v"(a0) : "v"(addr) : "memory");               \
        }                                                                             \
        __syncthreads();                                                              \
        out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x] + static_cast<float>(a0); \
    }
LDS_KERNEL(k_ds_write_b64, "ds_write_b64 %1, %0\nds_write_b64 %1, %0 offset:512\n", 0)
LDS_KERNEL(k_ds_write_b32, "ds_write_b32 %1, %1\nds_write_b32 %1, %1 offset:512\n", 0)
LDS_KERNEL(k_ds_read_b64, "ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\nds_read_b64 %0, %1 offset:512\ns_waitcnt lgkmcnt(0)\n", 0)

struct Case {
    const char* name;
    void (*fn)(float*, float);
    int per_iter;
};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1e3;  // kHz -> MHz
    std::printf("device %s: %d CUs, clock %.0f MHz\n", prop.name, cus, mhz);
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8 * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Case cases[] = {
        {"v_mul_f32", k_mul_f32, 32},       {"v_add_f32", k_add_f32, 32},       {"v_max3_f32", k_max3_f32, 32},
        {"v_med3_f32", k_med3_f32, 32},     {"v_cndmask_b32", k_cndmask, 32},   {"v_cmp_le_f32", k_cmp_f32, 32},
        {"v_fmac_f32 (VOP2)", k_fmac_f32, 32}, {"v_fma_f32 (VOP3)", k_fma_f32, 32}, {"v_max_f32 (VOP2)", k_max_f32, 32},
        {"v_cndmask e64 sgpr", k_cndmask_sgpr, 32}, {"v_cndmask vcc set", k_cndmask_vcc_set, 32},
        {"v_cndmask fresh dst", k_cndmask_fresh_dst, 32},
        {"v_bfi_b32", k_bfi, 32}, {"v_or_b32", k_or, 32}, {"v_ashrrev_i32", k_ashr, 32}, {"v_min_u32", k_min_u32, 32},
        {"v_min3_u32", k_min3_u32, 32}, {"v_add_u32", k_add_u32, 32}, {"v_lshl_add_u32", k_lshl_add, 32},

        {"v_mov_b32", k_mov_b32, 32},       {"v_pk_mul_f32", k_pk_mul_f32, 32}, {"v_pk_add_f32", k_pk_add_f32, 32},
        {"v_pk_fma_f32", k_pk_fma_f32, 32}, {"v_min_f64", k_min_f64, 32},       {"v_max_f64", k_max_f64, 32},
        {"v_add_f64", k_add_f64, 32},       {"v_cmp_lt_u64", k_cmp_u64, 32},    {"v_mov_b64", k_mov_b64, 32},
        {"v_pk_mov_b32", k_pk_mov_b32, 32}, {"s_add_u32", k_salu, 32},          {"v_mul+s_add 1:1", k_valu_salu, 32},
        {"ds_write2_b32", k_ds_write2, 16},
        {"ds_write2 exec=4", k_ds_write2_masked<4>, 16},
        {"ds_write2 exec=1", k_ds_write2_masked<1>, 16},
        {"ds_write2 exec=0", k_ds_write2_masked<0>, 16},
        {"ds_write_b64", k_ds_write_b64, 16},
        {"ds_write_b32", k_ds_write_b32, 16},
        {"ds_read_b64+wait", k_ds_read_b64, 16},
        {"accept seq (x1)", k_accept_seq, 8},
        {"accept seq no LDS", k_accept_seq_nolds, 8},
        {"candidate (x1)", k_candidate, 8},
        {"v_readlane_b32", k_readlane, 32},
        {"v_readlane + use x4", k_readlane_use, 32},
        {"v_writelane_b32", k_writelane, 32},
        {"v_mbcnt_lo", k_mbcnt, 32},
        {"sparse owner (x1)", k_sparse_owner, 8},
    };
    std::printf("%-18s %10s %10s %10s   (cycles per wave-instruction [or per sequence] per SIMD at 1, 2, 4, 5 waves/SIMD)\n", "instruction", "w=1", "w=2",
                "w=4 (w=5)");
    for (const Case& c : cases) {
        std::printf("%-18s", c.name);
        for (int w : {1, 2, 4, 5}) {
            const int grid = cus * 4 * w;
            c.fn<<<grid, 64>>>(out, 1.0f);  // warm-up
            CHECK(hipEventRecord(e0));
            c.fn<<<grid, 64>>>(out, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cycles = ms * 1e-3 * mhz * 1e6;
            const double instr_per_simd = static_cast<double>(ITER) * c.per_iter * w;
            std::printf(" %10.2f", cycles / instr_per_simd);
        }
        std::printf("\n");
    }
    CHECK(hipFree(out));
    return 0;
}
