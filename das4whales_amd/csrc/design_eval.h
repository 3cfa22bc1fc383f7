// Closed-form point evaluation of the reference's f-k mask designs on the fftshift-ed (k, f) grid -- shared by the
// dense design kernel (design.hip) and the fold that writes a design straight into a plan's pass-B order without a
// dense mask in between (fk_filter.hip, d4w_fk_set_mask_design_f32).  Reference: dsp.py:85-454, 457-702, 883-953.
#pragma once
#include "d4w_internal.h"

namespace d4w {

struct DesignArgs {
    int nx, ns;
    double kval, fval;            // 1/(nx*dk_spacing), 1/(ns*dt): axis value = (idx - n/2) * val
    double p[8];                  // speeds / band edges, meaning depends on the design
    int i0, i1;                   // half-open column range the reference's loops run over
    const double* hrow;           // [ns] band-pass row H(f) (hybrid_ninf: host-computed |H|^2)
};

__device__ __forceinline__ double axis_val(int idx, int n, double val) { return (double)(idx - n / 2) * val; }

constexpr double kHalfPi = 1.57079632679489661923;

// ---- classic speed fan, dsp.py:140-161 ---------------------------------------------------------
__device__ __forceinline__ double d_classic(const DesignArgs& A, int i, int j) {
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    const double cs_min = A.p[0], cp_min = A.p[1], cp_max = A.p[2], cs_max = A.p[3];
    const double s = fabs(f / k);
    double m = 1.0;
    if (s >= cs_min && s <= cp_min) m = sin(kHalfPi * (s - cs_min) / (cp_min - cs_min));
    if (s >= cp_max && s <= cs_max) m = 1.0 - sin(kHalfPi * (s - cp_max) / (cs_max - cp_max));
    if (s >= cs_max) m = 0.0;
    if (s < cs_min) m = 0.0;
    if (fabs(k) < 0.005) m = 0.0;
    return m;
}

// ---- hybrid: sine-tapered band x speed high-pass, before the flip (dsp.py:214-261) --------------
__device__ __forceinline__ double d_hybrid_core(const DesignArgs& A, int i, int j) {
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    const double cs_min = A.p[0], cp_min = A.p[1], fmin = A.p[2], fmax = A.p[3];
    const double fp_lo = fmin - 4.0, fp_hi = fmax + 4.0;
    double H = 0.0;
    if (f >= fp_lo && f <= fmin) H = sin(kHalfPi * (f - fp_lo) / (fmin - fp_lo));
    if (f >= fmin && f <= fmax) H = 1.0;
    if (f >= fmax && f <= fp_hi) H = cos(kHalfPi * (f - fmax) / (fmax - fp_hi));
    if (j < A.i0 || j >= A.i1) return H;
    const double ks = f / cs_min, kp = f / cp_min;
    double col = 0.0;
    if (ks != kp && k >= -ks && k <= -kp) col = -sin(kHalfPi * (k + ks) / (kp - ks));
    if (ks != kp && -k >= -ks && -k <= -kp) col = sin(kHalfPi * (k - ks) / (kp - ks));
    if (k < kp && k > -kp) col = 1.0;
    return H * col;
}

// ---- hybrid_ninf: Butterworth |H|^2 row x speed band-pass, before the flips (dsp.py:348-402) ----
__device__ __forceinline__ double d_ninf_core(const DesignArgs& A, int i, int j) {
    const double H = A.hrow[j];
    if (j < A.i0 || j >= A.i1 || H == 0.0) return H;     // (the negative-frequency half of the row is zero: H * col = 0)
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    const double cs_min = A.p[0], cp_min = A.p[1], cp_max = A.p[2], cs_max = A.p[3];
    const double ks_min = f / cs_max, kp_min = f / cp_max, ks_max = f / cs_min, kp_max = f / cp_min;
    double col = 0.0;
    if (ks_min != kp_min && k >= ks_min && k <= kp_min) col = sin(kHalfPi * (k - ks_min) / (kp_min - ks_min));
    if (ks_max != kp_max && k >= kp_max && k <= ks_max) col = -sin(kHalfPi * (k - ks_max) / (ks_max - kp_max));
    if (k > kp_min && k < kp_max) col = 1.0;
    return H * col;
}

// ---- box cores of the Gaussian designs ------------------------------------------------------------
__device__ __forceinline__ double d_gs_core(const DesignArgs& A, int i, int j) {       // dsp.py:508-536
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    const double cp_min = A.p[1], fmin = A.p[2], fmax = A.p[3];
    const double H = (f >= fmin && f <= fmax) ? 1.0 : 0.0;
    if (j < A.i0 || j >= A.i1) return H;
    const double kp = f / cp_min;
    return (k < kp && k > -kp) ? H : 0.0;
}
__device__ __forceinline__ double d_ninf_gs_core(const DesignArgs& A, int i, int j) {  // dsp.py:633-653
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    const double cp_min = A.p[1], cp_max = A.p[2], fmin = A.p[4], fmax = A.p[5];
    const double H = (f >= fmin && f <= fmax) ? 1.0 : 0.0;
    if (j < A.i0 || j >= A.i1) return H;
    return (k > -f / cp_min && k < -f / cp_max) ? H : 0.0;
}
__device__ __forceinline__ double d_wedge(const DesignArgs& A, int i, int j, double c) {  // dsp.py:930-931
    const double k = axis_val(i, A.nx, A.kval), f = axis_val(j, A.ns, A.fval);
    return (f < k * c && f < -k * c) ? 1.0 : 0.0;
}

// mode: 0 classic | 1 hybrid (C + fliplr C) | 2 hybrid_ninf (C + fliplr, + flipud) |
//       3 hybrid_gs pre-blur (C + fliplr C) | 4 hybrid_ninf_gs pre-blur (C) | 5 fk_filt wedge pre-blur
__device__ __forceinline__ double design_value(const DesignArgs& A, int mode, int i, int j) {
    const int ri = A.nx - 1 - i, rj = A.ns - 1 - j;
    switch (mode) {
        case 0: return d_classic(A, i, j);
        case 1: return d_hybrid_core(A, i, j) + d_hybrid_core(A, i, rj);
        case 2: return (d_ninf_core(A, i, j) + d_ninf_core(A, i, rj)) + (d_ninf_core(A, ri, j) + d_ninf_core(A, ri, rj));
        case 3: return d_gs_core(A, i, j) + d_gs_core(A, i, rj);
        case 4: return d_ninf_gs_core(A, i, j);
        default:
            return (d_wedge(A, i, j, A.p[0]) + d_wedge(A, i, rj, A.p[0])) - (d_wedge(A, i, j, A.p[1]) + d_wedge(A, i, rj, A.p[1]));
    }
}

// Hermitian part M_h = (M(k, f) + M(-k, -f)) / 2 at shifted-grid points (ip, jp) and its mirror (im, jm), each value
// rounded to float32 first (as the dense design kernel stores it).  The classic fan is even in (k, f) -- |f / k| and
// |k| are formed from exactly negated axis values -- so its mirror value is the same number: one evaluation.
__device__ __forceinline__ float design_folded(const DesignArgs& A, int mode, int ip, int jp, int im, int jm) {
    const float a = (float)design_value(A, mode, ip, jp);
    if (mode == 0) return a;
    return 0.5f * (a + (float)design_value(A, mode, im, jm));
}

inline DesignArgs make_design_args(int nx, int ns, double k_spacing, double t_spacing, const double* params8, int i0, int i1,
                                   const double* hrow_dev) {
    DesignArgs A;
    A.nx = nx; A.ns = ns;
    A.kval = 1.0 / ((double)nx * k_spacing);       // numpy.fft.fftfreq: val = 1.0 / (n * d)
    A.fval = 1.0 / ((double)ns * t_spacing);
    for (int i = 0; i < 8; ++i) A.p[i] = params8[i];
    A.i0 = i0; A.i1 = i1; A.hrow = hrow_dev;
    return A;
}

}  // namespace d4w
