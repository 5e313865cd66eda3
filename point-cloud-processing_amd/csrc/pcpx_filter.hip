// pcpx_filter.hip -- the consumers of sphere ranges for gfx950: the per-neighbour loops of
// pcp::algorithm::bilateral_filter_points / bilateral_filter_normals (include/pcp/algorithm/bilateral_filter.hpp) and
// pcp::algorithm::wlop::wlop (include/pcp/algorithm/wlop.hpp), fused into the range walk.  The reference materialises
// every range (kdtree.range_search -> std::vector of elements) and then loops over it; here a lane is one range centre,
// the wave walks the tree once for its 64 centres exactly as k_range does (pcpx_range.hip), and every point that passes a
// lane's d2 <= r*r test (sphere.hpp:52-56) is noted in a short per-lane LDS list that the lane folds into its accumulators
// every few leaves: no neighbour list ever reaches memory.  Per-point attributes the loops read (the neighbour's normal, its density weight) are gathered once into
// leaf order, next to the coordinates they belong to.
//
// Arithmetic is the reference's, in float, statement by statement (fp contraction off); what differs is the ORDER in which
// a centre's neighbours are summed (tree order here, kd-tree visiting order there: neither is specified by the reference's
// interface) and the last bit of expf.  Parity is therefore by tolerance, see tests/test_gpu_filters.py.
#include "pcpx_device.h"

namespace pcpx {

namespace {

// One record per leaf slot, in leaf order: the point and the per-point attribute its range loop reads (the neighbour's
// normal; its density weight), together in one 32-byte sector so that a lane working its list off fetches a neighbour
// with one or two wide loads instead of six scattered dwords (the texture addresser, shared by the CU's four SIMDs, was
// the limiter when coordinates and attributes were fetched from the leaves and from SoA attribute arrays).
struct __attribute__((aligned(16))) Rec4 {  // x y z + one scalar
    float x, y, z, a;
};
struct __attribute__((aligned(32))) Rec8 {  // x y z + a 3-vector
    float x, y, z, a, b, c, pad0, pad1;
};
static_assert(sizeof(Rec4) == 16 && sizeof(Rec8) == 32, "record sizes");

// C = 3: Rec8 with the attribute rows (n_in x 3, input order); C = 1: Rec4 with a scalar per point; C = 0: Rec4, a = 0.
// Padding slots hold NaN coordinates (they are never listed) and zero attributes.
template <int C>
__global__ __launch_bounds__(256) void k_leaf_records(const Leaf* __restrict__ leaves, u32 nslots, const float* __restrict__ attr,
                                                      float* __restrict__ out)
{
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nslots) return;
    const Leaf& lf = leaves[p / LEAF];
    const u32 s = p % LEAF;
    const u32 id = lf.id[s];
    const bool real = id != INVALID_ID;
    if (C == 3) {
        Rec8 r;
        r.x = lf.x[s], r.y = lf.y[s], r.z = lf.z[s];
        r.a = real ? attr[3ull * id] : 0.f;
        r.b = real ? attr[3ull * id + 1] : 0.f;
        r.c = real ? attr[3ull * id + 2] : 0.f;
        r.pad0 = r.pad1 = 0.f;
        reinterpret_cast<Rec8*>(out)[p] = r;
    } else {
        Rec4 r;
        r.x = lf.x[s], r.y = lf.y[s], r.z = lf.z[s];
        r.a = (C == 1 && real) ? attr[id] : 0.f;
        reinterpret_cast<Rec4*>(out)[p] = r;
    }
}

__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ p, u64 n, float v)
{
    const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// x[i] = xyz[sample[i]] (wlop.hpp:337-342); an index outside the cloud gives a NaN row
__global__ __launch_bounds__(256) void k_take_rows(const float* __restrict__ xyz, u64 n, const u64* __restrict__ sample, u64 m,
                                                   float* __restrict__ out)
{
    const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const u64 j = sample[i];
    const float nan = __builtin_nanf("");
    out[3 * i] = j < n ? xyz[3 * j] : nan;
    out[3 * i + 1] = j < n ? xyz[3 * j + 1] : nan;
    out[3 * i + 2] = j < n ? xyz[3 * j + 2] : nan;
}

// ------------------------------------------------------------------------------------------------
// the walk: one group of 64 range centres, one per lane (self: the centres are the indexed points themselves, in leaf
// order; batch: prepared queries).  The wave-uniform walk tests a leaf's 8 points against all 64 ranges (8 VALU each); an
// accepted point is only NOTED -- its leaf slot appended to the lane's own column of an LDS list -- because the
// arithmetic of one neighbour (two exp, two sqrt, three divisions for the bilateral weight; far more for its Jacobian)
// run under the EXEC mask of the few lanes that accepted that very point would keep ~5 % of the lanes busy.  When a
// column could overflow, and at the end, every lane works its own list off, all lanes busy: slot -> the point's record (one
// 16- or 32-byte load per lane), then Acc::add.  A lane's neighbours are still added in
// the order the walk met them, so the result does not depend on where the flushes fall.
// Acc supplies begin / add (one neighbour) / finish.
// ------------------------------------------------------------------------------------------------
// A list entry is 16 bits: the slot relative to the first slot of the leaf the walk stood at when the lists were last
// emptied (the walk yields leaves in ascending order, so the offset is never negative; the lists are emptied before it
// could exceed 16 bits).  Half the LDS of 32-bit entries, and LDS is what bounds the resident waves here -- the walk hides
// its scalar-load latency with them (measured: 32-bit entries at 32 per lane 7.5 ms for the bilateral points loop over
// 10 M ranges of ~42 points, 16-bit entries 5.7 ms).
// Acc::CAP = entries per lane between two flushes, per loop, by measurement: a long list keeps more lanes busy when it
// is worked off (the normal loop's Jacobian, ~400 instructions per neighbour: 13.0 ms at 32 entries, 8.0 ms at 64), a
// short one leaves LDS for more waves (everything else: within 3 % between 32 and 48).
#ifndef PCPX_CAP_POINTS
#define PCPX_CAP_POINTS 40
#endif
#ifndef PCPX_CAP_NORMALS
#define PCPX_CAP_NORMALS 64
#endif
#ifndef PCPX_CAP_DENSITY
#define PCPX_CAP_DENSITY 40
#endif
#ifndef PCPX_CAP_MEDIAN
#define PCPX_CAP_MEDIAN 40
#endif
#ifndef PCPX_CAP_REPULSION
#define PCPX_CAP_REPULSION 40
#endif
typedef unsigned short list_entry;
constexpr u32 LIST_SPAN = 65536u / LEAF;  // leaves one epoch of the lists can address

template <class Acc>
__device__ __forceinline__ void flush_list(const list_entry* list, const u32 base_slot, const u32 lane, const u32 cnt, const float qx,
                                           const float qy, const float qz, Acc& acc)
{
    for (u32 i = 0; any_lane(i < cnt); ++i) {
        if (i < cnt) {
            const typename Acc::Rec r = acc.recs[base_slot + list[i * GROUP + lane]];
            const float dx = r.x - qx, dy = r.y - qy, dz = r.z - qz;
            acc.add(r, sq3(dx, dy, dz));  // (the same d2 the walk computed)
        }
    }
}

template <bool SELF, class Acc>
__device__ __forceinline__ void visit_group(const TreeView& t, const QueryView& qv, const u32 g, const float radius, Acc& acc,
                                            list_entry* list, const u32 lane)
{
    const u32 p = g * GROUP + lane;
    const u32 nq = SELF ? t.n : qv.nq;
    const bool valid = p < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    u32 row = 0;
    if (valid) {
        if (SELF) {
            const Leaf& lf = t.leaves[p / LEAF];
            qx = lf.x[p % LEAF];
            qy = lf.y[p % LEAF];
            qz = lf.z[p % LEAF];
            row = lf.id[p % LEAF];
        } else {
            qx = qv.qx[p];
            qy = qv.qy[p];
            qz = qv.qz[p];
            row = qv.row[p];
        }
    }
    const float r2 = valid ? radius * radius : -1.f;  // sphere.hpp:55 radius * radius in float; -1: idle lane
    acc.begin(qx, qy, qz, p, valid);
    auto need = [&](const NodeBox& b) { return box_d2(b, qx, qy, qz) <= r2; };
    Walker wk;
    u32 unit = 0, nexp = 0, cnt = 0;  // (the walk yields UNITS of the tree's bottom level: UNIT_LEAVES consecutive leaf records each)
    bool more = wk.start(t, need, nexp);
    if (!more) more = wk.next(t, need, unit, nexp);
    u32 base_leaf = unit * UNIT_LEAVES;  // wave-uniform: the epoch of the lists
    while (more) {
      for (u32 leaf = unit * UNIT_LEAVES; leaf < (unit + 1u) * UNIT_LEAVES && leaf < t.nleaves; ++leaf) {
        if (leaf - base_leaf >= LIST_SPAN) {  // offsets from here on would not fit an entry
            flush_list(list, base_leaf * LEAF, lane, cnt, qx, qy, qz, acc);
            cnt = 0;
            base_leaf = leaf;
        }
        const Leaf lf = load_const(t.leaves + leaf);
        const u32 rel = (leaf - base_leaf) * LEAF;
#pragma unroll
        for (int j = 0; j < LEAF; ++j) {
            const float dx = lf.x[j] - qx, dy = lf.y[j] - qy, dz = lf.z[j] - qz;
            if (sq3(dx, dy, dz) <= r2) {  // = common::squared_distance(centre, point) <= r*r, norm.hpp:102-112
                list[cnt * GROUP + lane] = static_cast<list_entry>(rel + j);
                ++cnt;
            }
        }
        if (any_lane(cnt > Acc::CAP - LEAF)) {  // the next leaf could overflow a column
            flush_list(list, base_leaf * LEAF, lane, cnt, qx, qy, qz, acc);
            cnt = 0;
            base_leaf = leaf;
        }
      }
        more = wk.next(t, need, unit, nexp);
    }
    flush_list(list, base_leaf * LEAF, lane, cnt, qx, qy, qz, acc);
    if (valid) acc.finish(row);
}

// ---- bilateral filter --------------------------------------------------------------------------

// gaussian / dgaussian of bilateral_filter.hpp:361-370, :528-537 with their sigma-only factors hoisted to the host
// (same float expressions): g(r) = coeff * exp(-(r*r) / two_s2), dg(r) = (-r / dcoeff_den) * exp(-(r*r) / two_s2)
struct Gauss {
    float two_s2;      // 2 * (sigma * sigma)
    float coeff;       // 1 / (sigma * sqrt(2 * pi))
    float dcoeff_den;  // (sigma * (sigma * sigma)) * sqrt(2 * pi)
    __device__ __forceinline__ float g(float r) const
    {
        const float r2 = r * r;
        return coeff * expf(-r2 / two_s2);
    }
    __device__ __forceinline__ float dg(float r) const
    {
        const float r2 = r * r;
        return (-r / dcoeff_den) * expf(-r2 / two_s2);
    }
};

__device__ __forceinline__ float norm3(float x, float y, float z)
{
    const float xx = x * x, yy = y * y, zz = z * z;
    return sqrtf(xx + yy + zz);  // norm.hpp:47-66
}

// bilateral::detail::compute_pi (bilateral_filter.hpp:47-101)
struct BilateralPoints {
    using Rec = Rec8;
    static constexpr int CAP = PCPX_CAP_POINTS;
    const Rec8* recs;  // point + its normal
    Gauss f, g;
    float* out;  // n_in x 3
    float sx, sy, sz, k, ax, ay, az;
    __device__ __forceinline__ void begin(float qx, float qy, float qz, u32, bool)
    {
        sx = qx, sy = qy, sz = qz;
        k = ax = ay = az = 0.f;
    }
    __device__ __forceinline__ void add(const Rec8& r, float)
    {
        const float px = r.x, py = r.y, pz = r.z, nx = r.a, ny = r.b, nz = r.c;
        // projection (:372-379): s + inner_product(p - s, n) * n
        const float spx = px - sx, spy = py - sy, spz = pz - sz;
        const float xx = nx * spx, yy = ny * spy, zz = nz * spz;
        const float d = xx + yy + zz;
        const float prx = sx + d * nx, pry = sy + d * ny, prz = sz + d * nz;
        const float rf = norm3(sx - px, sy - py, sz - pz);
        const float rg = norm3(prx - sx, pry - sy, prz - sz);
        const float w = f.g(rf) * g.g(rg);
        k += w;
        ax += w * prx;
        ay += w * pry;
        az += w * prz;
    }
    __device__ __forceinline__ void finish(u32 row)
    {
        out[3ull * row] = ax / k;
        out[3ull * row + 1] = ay / k;
        out[3ull * row + 2] = az / k;
    }
};

// Eigen 3.3.8 reduces a fixed 3-vector as a0 + (a1 + a2), and normalized() / normalize() divide only when the squared
// norm is > 0 (Eigen/src/Core/Redux.h redux_novec_unroller, Eigen/src/Core/Dot.h)
__device__ __forceinline__ float sum3(float a, float b, float c) { return a + (b + c); }
__device__ __forceinline__ void normalized3(float x, float y, float z, float& ox, float& oy, float& oz)
{
    const float s = sum3(x * x, y * y, z * z);
    if (s > 0.f) {
        const float n = sqrtf(s);
        ox = x / n, oy = y / n, oz = z / n;
    } else {
        ox = x, oy = y, oz = z;
    }
}

// bilateral::detail::compute_ni (bilateral_filter.hpp:103-269): the Jacobian of the filter at s applied to s's normal
struct BilateralNormals {
    using Rec = Rec8;
    static constexpr int CAP = PCPX_CAP_NORMALS;
    const Rec8* recs;  // point + its normal
    Gauss f, g;
    float* out;  // n_in x 3
    float sx, sy, sz, nsx, nsy, nsz;
    float k, pi0, pi1, pi2, gk0, gk1, gk2;
    float J00, J01, J02, J10, J11, J12, J20, J21, J22;
    __device__ __forceinline__ void begin(float qx, float qy, float qz, u32 p, bool valid)
    {
        sx = qx, sy = qy, sz = qz;
        nsx = nsy = nsz = 0.f;
        if (valid) {  // the centre's own normal: its own record
            const Rec8 mine = recs[p];
            nsx = mine.a, nsy = mine.b, nsz = mine.c;
        }
        k = pi0 = pi1 = pi2 = gk0 = gk1 = gk2 = 0.f;
        J00 = J01 = J02 = J10 = J11 = J12 = J20 = J21 = J22 = 0.f;
    }
    __device__ __forceinline__ void add(const Rec8& r, float)
    {
        const float px = r.x, py = r.y, pz = r.z, nx = r.a, ny = r.b, nz = r.c;
        const float spx = px - sx, spy = py - sy, spz = pz - sz;
        const float xx = nx * spx, yy = ny * spy, zz = nz * spz;
        const float d = xx + yy + zz;
        const float prx = sx + d * nx, pry = sy + d * ny, prz = sz + d * nz;
        const float s0 = sx - px, s1 = sy - py, s2 = sz - pz;        // sp  = s - p
        const float q0 = prx - sx, q1 = pry - sy, q2 = prz - sz;     // sps = projection(s) - s
        const float rf = sqrtf(sum3(s0 * s0, s1 * s1, s2 * s2));
        const float rg = sqrtf(sum3(q0 * q0, q1 * q1, q2 * q2));
        const float wf = f.g(rf), wg = g.g(rg);
        const float w = wf * wg;
        k += w;
        pi0 += w * prx;
        pi1 += w * pry;
        pi2 += w * prz;
        const float wdf = f.dg(rf);
        float u0, u1, u2;
        normalized3(s0, s1, s2, u0, u1, u2);
        const float gf0 = u0 * wdf, gf1 = u1 * wdf, gf2 = u2 * wdf;
        // Jacobian of the projection as written at :213-222
        const float P00 = 1.f - (nx * nx), P11 = 1.f - (ny * ny), P22 = 1.f - (nz * nz);
        const float P01 = nx * ny, P02 = nx * nz, P12 = ny * nz;
        const float wdg = g.dg(rg);
        float v0, v1, v2;
        normalized3(q0, q1, q2, v0, v1, v2);
        const float gg0 = (sum3(v0 * P00, v1 * P01, v2 * P02) - v0) * wdg;
        const float gg1 = (sum3(v0 * P01, v1 * P11, v2 * P12) - v1) * wdg;
        const float gg2 = (sum3(v0 * P02, v1 * P12, v2 * P22) - v2) * wdg;
        gk0 += (gf0 * wg) + (wf * gg0);
        gk1 += (gf1 * wg) + (wf * gg1);
        gk2 += (gf2 * wg) + (wf * gg2);
        // J += (Jpi * wf * wg) + (sps * grad_f * wg) + (sps * wf * grad_g)
        const float q0f = q0 * wf, q1f = q1 * wf, q2f = q2 * wf;
        J00 += ((P00 * wf) * wg + (q0 * gf0) * wg) + q0f * gg0;
        J01 += ((P01 * wf) * wg + (q0 * gf1) * wg) + q0f * gg1;
        J02 += ((P02 * wf) * wg + (q0 * gf2) * wg) + q0f * gg2;
        J10 += ((P01 * wf) * wg + (q1 * gf0) * wg) + q1f * gg0;
        J11 += ((P11 * wf) * wg + (q1 * gf1) * wg) + q1f * gg1;
        J12 += ((P12 * wf) * wg + (q1 * gf2) * wg) + q1f * gg2;
        J20 += ((P02 * wf) * wg + (q2 * gf0) * wg) + q2f * gg0;
        J21 += ((P12 * wf) * wg + (q2 * gf1) * wg) + q2f * gg1;
        J22 += ((P22 * wf) * wg + (q2 * gf2) * wg) + q2f * gg2;
    }
    __device__ __forceinline__ void finish(u32 row)
    {
        // quotient rule (:248-249): J = (1 / k^2) * (J_pi_f_g * k - pi_f_g * grad_k)
        const float inv = 1.f / (k * k);
        const float a00 = inv * (J00 * k - pi0 * gk0), a01 = inv * (J01 * k - pi0 * gk1), a02 = inv * (J02 * k - pi0 * gk2);
        const float a10 = inv * (J10 * k - pi1 * gk0), a11 = inv * (J11 * k - pi1 * gk1), a12 = inv * (J12 * k - pi1 * gk2);
        const float a20 = inv * (J20 * k - pi2 * gk0), a21 = inv * (J21 * k - pi2 * gk1), a22 = inv * (J22 * k - pi2 * gk2);
        const float x = sum3(a00 * nsx, a01 * nsy, a02 * nsz);
        const float y = sum3(a10 * nsx, a11 * nsy, a12 * nsz);
        const float z = sum3(a20 * nsx, a21 * nsy, a22 * nsz);
        float ox, oy, oz;
        normalized3(x, y, z, ox, oy, oz);
        out[3ull * row] = ox;
        out[3ull * row + 1] = oy;
        out[3ull * row + 2] = oz;
    }
};

// ---- WLOP ----------------------------------------------------------------------------------------

constexpr float WLOP_EPS = 1e-9f;  // wlop.hpp:45,85,131,195: eps of every floating_point_equals there, in scalar_type
__device__ __forceinline__ bool near_eq(float a, float b) { return fabsf(a - b) < WLOP_EPS; }  // vector3d_queries.hpp:30-35

// compute_vj / compute_wi (wlop.hpp:29-105): 1 + sum of theta(r2) over the range, points equal to the centre skipped
struct WlopDensity {
    using Rec = Rec4;
    static constexpr int CAP = PCPX_CAP_DENSITY;
    const Rec4* recs;
    float h16;   // (h * h) / 16
    float* out;  // n_in
    float sx, sy, sz, v;
    __device__ __forceinline__ void begin(float qx, float qy, float qz, u32, bool)
    {
        sx = qx, sy = qy, sz = qz;
        v = 1.f;
    }
    __device__ __forceinline__ void add(const Rec4& r, float d2)
    {
        if (near_eq(sx, r.x) && near_eq(sy, r.y) && near_eq(sz, r.z)) return;
        v += expf(-d2 / h16);
    }
    __device__ __forceinline__ void finish(u32 row) { out[row] = v; }
};

// solve_first_energy_median (wlop.hpp:107-170): centres are the samples x, the tree holds the input cloud and its v_j
struct WlopMedian {
    using Rec = Rec4;
    static constexpr int CAP = PCPX_CAP_MEDIAN;
    const Rec4* recs;  // cloud point + its v_j
    float h16;
    float* out;  // I x 3
    float sx, sy, sz, sum, mx, my, mz;
    __device__ __forceinline__ void begin(float qx, float qy, float qz, u32, bool)
    {
        sx = qx, sy = qy, sz = qz;
        sum = mx = my = mz = 0.f;
    }
    __device__ __forceinline__ void add(const Rec4& rec, float d2)
    {
        const float px = rec.x, py = rec.y, pz = rec.z;
        if (near_eq(sx, px) && near_eq(sy, py) && near_eq(sz, pz)) return;
        const float r = sqrtf(d2);
        const float v = rec.a;
        const float alpha = near_eq(r, 0.f) ? 0.f : expf(-d2 / h16) / r;
        const float coeff = near_eq(v, 0.f) ? 0.f : alpha / v;
        mx += coeff * px;
        my += coeff * py;
        mz += coeff * pz;
        sum += coeff;
    }
    __device__ __forceinline__ void finish(u32 row)
    {
        const bool none = near_eq(sum, 0.f);
        out[3ull * row] = none ? sx : mx / sum;
        out[3ull * row + 1] = none ? sy : my / sum;
        out[3ull * row + 2] = none ? sz : mz / sum;
    }
};

// solve_second_energy_repulsion_force (wlop.hpp:172-229) + the update x' = median + repulsion (:406-409): centres and
// tree are both the samples x
struct WlopRepulsion {
    using Rec = Rec4;
    static constexpr int CAP = PCPX_CAP_REPULSION;
    const Rec4* recs;  // sample + its w_i
    float h16, mu;
    const float* median;  // I x 3
    float* out;           // I x 3
    float sx, sy, sz, sum, rx, ry, rz;
    __device__ __forceinline__ void begin(float qx, float qy, float qz, u32, bool)
    {
        sx = qx, sy = qy, sz = qz;
        sum = rx = ry = rz = 0.f;
    }
    __device__ __forceinline__ void add(const Rec4& rec, float d2)
    {
        const float px = rec.x, py = rec.y, pz = rec.z;
        if (near_eq(px, sx) && near_eq(py, sy) && near_eq(pz, sz)) return;
        const float dx = sx - px, dy = sy - py, dz = sz - pz;
        const float r = sqrtf(d2);
        const float beta = near_eq(r, 0.f) ? 0.f : expf(-d2 / h16) / r;
        const float coeff = rec.a * beta;
        rx += coeff * dx;
        ry += coeff * dy;
        rz += coeff * dz;
        sum += coeff;
    }
    __device__ __forceinline__ void finish(u32 row)
    {
        const float fct = near_eq(sum, 0.f) ? 0.f : mu / sum;
        out[3ull * row] = median[3ull * row] + fct * rx;
        out[3ull * row + 1] = median[3ull * row + 1] + fct * ry;
        out[3ull * row + 2] = median[3ull * row + 2] + fct * rz;
    }
};

// one single-wave workgroup per group of 64 centres, XCD-aware block order (as k_range)
template <bool SELF, class Acc>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_range_accumulate(TreeView t, QueryView qv, u32 group_end, float radius, Acc acc)
{
    __shared__ list_entry lists[WAVES_PER_BLOCK][Acc::CAP * GROUP];
    const u32 lane = threadIdx.x & 63u;
    const u32 g = virtual_block() * WAVES_PER_BLOCK + wave_in_block();
    if (g >= group_end) return;
    visit_group<SELF>(t, qv, g, radius, acc, lists[wave_in_block()], lane);
}

template <bool SELF, class Acc>
int launch_accumulate(Index& ix, const QueryView& qv, float radius, const Acc& acc, const char* what)
{
    const u64 centres = SELF ? ix.n : qv.nq;
    const u64 groups = (centres + GROUP - 1) / GROUP;
    if (groups == 0) return PCPX_OK;
    ProfileScope prof(ix, PCPX_K_RANGE);
    k_range_accumulate<SELF, Acc><<<grid_for_groups(groups), 64 * WAVES_PER_BLOCK, 0, ix.stream>>>(ix.view(), qv, static_cast<u32>(groups), radius, acc);
    return check_hip(hipGetLastError(), what, __FILE__, __LINE__);
}

Gauss gauss_of(float sigma)
{
    const float pi = static_cast<float>(3.14159265358979323846);
    const float s2 = sigma * sigma;
    const float s3 = sigma * s2;
    Gauss gs;
    gs.two_s2 = 2 * s2;
    gs.coeff = 1.f / (sigma * std::sqrt(2.f * pi));
    gs.dcoeff_den = s3 * std::sqrt(2.f * pi);
    return gs;
}

inline float h_over_4_squared(float h)
{
    const float h2 = h * h;
    return h2 / 16.f;  // wlop.hpp:317-319
}

}  // namespace

size_t leaf_record_bytes(u64 points, int components)
{
    return static_cast<size_t>((points + LEAF - 1) / LEAF) * LEAF * (components == 3 ? sizeof(Rec8) : sizeof(Rec4));
}

int launch_leaf_records(Index& ix, const float* d_attr, int components, void* d_records)
{
    const u32 nslots = ix.nleaves * LEAF;
    if (nslots == 0) return PCPX_OK;
    const u32 blocks = (nslots + 255) / 256;
    float* out = static_cast<float*>(d_records);
    if (components == 3)
        k_leaf_records<3><<<blocks, 256, 0, ix.stream>>>(ix.d_leaves, nslots, d_attr, out);
    else if (components == 1)
        k_leaf_records<1><<<blocks, 256, 0, ix.stream>>>(ix.d_leaves, nslots, d_attr, out);
    else if (components == 0)
        k_leaf_records<0><<<blocks, 256, 0, ix.stream>>>(ix.d_leaves, nslots, nullptr, out);
    else
        return PCPX_ERR_INVALID;
    return check_hip(hipGetLastError(), "k_leaf_records launch", __FILE__, __LINE__);
}

int launch_fill_f32(float* d_p, u64 n, float v, hipStream_t s)
{
    if (n == 0) return PCPX_OK;
    k_fill_f32<<<static_cast<u32>((n + 255) / 256), 256, 0, s>>>(d_p, n, v);
    return check_hip(hipGetLastError(), "k_fill_f32 launch", __FILE__, __LINE__);
}

int launch_take_rows(const float* d_xyz, u64 n, const u64* d_sample, u64 m, float* d_out, hipStream_t s)
{
    if (m == 0) return PCPX_OK;
    k_take_rows<<<static_cast<u32>((m + 255) / 256), 256, 0, s>>>(d_xyz, n, d_sample, m, d_out);
    return check_hip(hipGetLastError(), "k_take_rows launch", __FILE__, __LINE__);
}

int launch_bilateral(Index& ix, const void* d_records, float sigmaf, float sigmag, bool normals_mode, float* d_out)
{
    const float radius = 2.f * sigmaf;  // bilateral_filter.hpp:73 / :143
    const QueryView none{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    const Rec8* nrm = static_cast<const Rec8*>(d_records);
    if (normals_mode) {
        BilateralNormals acc{};
        acc.recs = nrm;
        acc.f = gauss_of(sigmaf);
        acc.g = gauss_of(sigmag);
        acc.out = d_out;
        return launch_accumulate<true>(ix, none, radius, acc, "bilateral normals launch");
    }
    BilateralPoints acc{};
    acc.recs = nrm;
    acc.f = gauss_of(sigmaf);
    acc.g = gauss_of(sigmag);
    acc.out = d_out;
    return launch_accumulate<true>(ix, none, radius, acc, "bilateral points launch");
}

int launch_wlop_density(Index& ix, float h, const void* d_records, float* d_out)
{
    const QueryView none{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    WlopDensity acc{};
    acc.recs = static_cast<const Rec4*>(d_records);
    acc.h16 = h_over_4_squared(h);
    acc.out = d_out;
    return launch_accumulate<true>(ix, none, h, acc, "wlop density launch");
}

int launch_wlop_median(Index& cloud, const QueryView& samples, float h, const void* d_records_vj, float* d_median)
{
    WlopMedian acc{};
    acc.recs = static_cast<const Rec4*>(d_records_vj);
    acc.h16 = h_over_4_squared(h);
    acc.out = d_median;
    return launch_accumulate<false>(cloud, samples, h, acc, "wlop median launch");
}

int launch_wlop_repulsion(Index& samples, float h, float mu, const void* d_records_wi, const float* d_median, float* d_out)
{
    const QueryView none{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    WlopRepulsion acc{};
    acc.recs = static_cast<const Rec4*>(d_records_wi);
    acc.h16 = h_over_4_squared(h);
    acc.mu = mu;
    acc.median = d_median;
    acc.out = d_out;
    return launch_accumulate<true>(samples, none, h, acc, "wlop repulsion launch");
}

}  // namespace pcpx
