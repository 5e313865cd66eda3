// pcpx_eig3.h -- PCA normal of a 3x3 scatter matrix on the device (shared by the fused kNN kernel and k_normals).
#ifndef PCPX_EIG3_H
#define PCPX_EIG3_H

#include "pcpx_device.h"

namespace pcpx {
namespace {

// ------------------------------------------------------------------------------------------------
// PCA normal of one neighbourhood per thread.
// Restates pcp::estimate_normal (include/pcp/common/normals/normal_estimation.hpp:41-77): row mean,
// centred scatter matrix V'V'^T (not divided by n), Eigen 3.3.8 SelfAdjointEigenSolver<Matrix3f>
// ::compute (scale, closed-form 3x3 tridiagonalisation, implicit-shift QL, ascending sort), column of
// the smallest eigenvalue with the reference's "last tie wins" ifs.  All in float32 without FMA, in
// the same operation order as the CPU restatement used by the tests, so results are bit-comparable with it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void make_givens(float p, float q, float& c, float& s)
{
    if (q == 0.f) {
        c = p < 0.f ? -1.f : 1.f;
        s = 0.f;
    } else if (p == 0.f) {
        c = 0.f;
        s = q < 0.f ? 1.f : -1.f;
    } else if (fabsf(p) > fabsf(q)) {
        float t = q / p;
        float u = sqrtf(1.f + t * t);
        if (p < 0.f) u = -u;
        c = 1.f / u;
        s = -t * c;
    } else {
        float t = p / q;
        float u = sqrtf(1.f + t * t);
        if (q < 0.f) u = -u;
        s = -1.f / u;
        c = -t * s;
    }
}

__device__ __forceinline__ float eig_hypot(float x, float y)
{
    float ax = fabsf(x), ay = fabsf(y);
    float p, qp;
    if (ax > ay) {
        p = ax;
        qp = ay / p;
    } else {
        p = ay;
        qp = ax / p;
    }
    if (p == 0.f) return 0.f;
    return p * sqrtf(1.f + qp * qp);
}

__device__ void eig3_smallest(float a00, float a10, float a20, float a11, float a21, float a22, float normal[3],
                              float evals[3])
{
    float scale = fmaxf(fmaxf(fmaxf(fabsf(a00), fabsf(a10)), fmaxf(fabsf(a20), fabsf(a11))),
                        fmaxf(fabsf(a21), fabsf(a22)));
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a20 /= scale; a11 /= scale; a21 /= scale; a22 /= scale;
    float d0, d1, d2, e0, e1;
    float q00 = 1.f, q10 = 0.f, q20 = 0.f, q01 = 0.f, q11 = 1.f, q21 = 0.f, q02 = 0.f, q12 = 0.f, q22 = 1.f;
    const float tiny = std::numeric_limits<float>::min();
    d0 = a00;
    float v1norm2 = a20 * a20;
    if (v1norm2 <= tiny) {
        d1 = a11; d2 = a22; e0 = a10; e1 = a21;
    } else {
        float beta = sqrtf(a10 * a10 + v1norm2);
        float inv_beta = 1.f / beta;
        float m01 = a10 * inv_beta, m02 = a20 * inv_beta;
        float q = 2.f * m01 * a21 + m02 * (a22 - a11);
        d1 = a11 + m02 * q;
        d2 = a22 - m02 * q;
        e0 = beta;
        e1 = a21 - m01 * q;
        q11 = m01; q21 = m02; q12 = m02; q22 = -m01;
    }
    // registers instead of arrays: diag = {d0,d1,d2}, sub = {e0,e1}, Q columns {q?0,q?1,q?2}
    const float precision = 2.f * std::numeric_limits<float>::epsilon();
    int end = 2, start = 0, iter = 0;
    bool converged = true;
    auto rot_cols01 = [&](float c, float s) {
        float x, y;
        x = q00; y = q01; q00 = c * x - s * y; q01 = s * x + c * y;
        x = q10; y = q11; q10 = c * x - s * y; q11 = s * x + c * y;
        x = q20; y = q21; q20 = c * x - s * y; q21 = s * x + c * y;
    };
    auto rot_cols12 = [&](float c, float s) {
        float x, y;
        x = q01; y = q02; q01 = c * x - s * y; q02 = s * x + c * y;
        x = q11; y = q12; q11 = c * x - s * y; q12 = s * x + c * y;
        x = q21; y = q22; q21 = c * x - s * y; q22 = s * x + c * y;
    };
    while (end > 0) {
        if (start <= 0 && 0 < end)
            if (fabsf(e0) <= (fabsf(d0) + fabsf(d1)) * precision || fabsf(e0) <= tiny) e0 = 0.f;
        if (start <= 1 && 1 < end)
            if (fabsf(e1) <= (fabsf(d1) + fabsf(d2)) * precision || fabsf(e1) <= tiny) e1 = 0.f;
        while (end > 0 && (end == 2 ? e1 : e0) == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 90) { converged = false; break; }
        start = end - 1;
        while (start > 0 && (start == 1 ? e0 : 0.f) != 0.f) start--;
        // tridiagonal_qr_step(start, end)
        float dem1 = end == 2 ? d1 : d0, de = end == 2 ? d2 : d1, ee = end == 2 ? e1 : e0;
        float td = (dem1 - de) * 0.5f;
        float mu = de;
        if (td == 0.f) {
            mu -= fabsf(ee);
        } else {
            float e2 = ee * ee;
            float h = eig_hypot(td, ee);
            if (e2 == 0.f) mu -= (ee / (td + (td > 0.f ? 1.f : -1.f))) * (ee / h);
            else mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = (start == 0 ? d0 : d1) - mu;
        float z = start == 0 ? e0 : e1;
        for (int k = start; k < end; ++k) {
            float c, s;
            make_givens(x, z, c, s);
            float dk = k == 0 ? d0 : d1, dk1 = k == 0 ? d1 : d2, sk = k == 0 ? e0 : e1;
            float sdk = s * dk + c * sk;
            float dkp1 = s * sk + c * dk1;
            float ndk = c * (c * dk - s * sk) - s * (c * sk - s * dk1);
            float ndk1 = s * sdk + c * dkp1;
            float nsk = c * sdk - s * dkp1;
            if (k == 0) { d0 = ndk; d1 = ndk1; e0 = nsk; }
            else { d1 = ndk; d2 = ndk1; e1 = nsk; }
            if (k > start) e0 = c * e0 - s * z;  // k == 1, start == 0: sub[k-1] = sub[0]
            x = nsk;
            if (k < end - 1) {  // k == 0, end == 2
                z = -s * e1;
                e1 = c * e1;
            }
            if (k == 0) rot_cols01(c, s);
            else rot_cols12(c, s);
        }
    }
    if (converged) {
        // ascending selection sort (first minimum wins), swapping eigenvector columns
        auto swap01 = [&]() {
            float t;
            t = d0; d0 = d1; d1 = t;
            t = q00; q00 = q01; q01 = t; t = q10; q10 = q11; q11 = t; t = q20; q20 = q21; q21 = t;
        };
        auto swap02 = [&]() {
            float t;
            t = d0; d0 = d2; d2 = t;
            t = q00; q00 = q02; q02 = t; t = q10; q10 = q12; q12 = t; t = q20; q20 = q22; q22 = t;
        };
        auto swap12 = [&]() {
            float t;
            t = d1; d1 = d2; d2 = t;
            t = q01; q01 = q02; q02 = t; t = q11; q11 = q12; q12 = t; t = q21; q21 = q22; q22 = t;
        };
        int kmin = 0;
        float mv = d0;
        if (d1 < mv) { mv = d1; kmin = 1; }
        if (d2 < mv) { mv = d2; kmin = 2; }
        if (kmin == 1) swap01();
        else if (kmin == 2) swap02();
        if (d2 < d1) swap12();
    }
    float l0 = d0 * scale, l1 = d1 * scale, l2 = d2 * scale;
    evals[0] = l0; evals[1] = l1; evals[2] = l2;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (l0 <= l1 && l0 <= l2) { nx = q00; ny = q10; nz = q20; }
    if (l1 <= l0 && l1 <= l2) { nx = q01; ny = q11; nz = q21; }
    if (l2 <= l0 && l2 <= l1) { nx = q02; ny = q12; nz = q22; }
    normal[0] = nx; normal[1] = ny; normal[2] = nz;
}

}  // namespace
}  // namespace pcpx

#endif
