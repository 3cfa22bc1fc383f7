// Types shared by the f-k filter's planner (fk_filter.hip) and by shape configurations compiled on demand
// (das4whales_amd/fkjit.py): the kernel argument blocks, the pass kernels of fk_fast.h and the table entry that
// names one instantiated shape.
#pragma once
#include "fft_lds.h"

namespace d4w {

struct FkDims {
    int nx, ns, M, C1, C2, N1, N2, TA, TC;
};

struct FkDev {  // kernel argument block (by value)
    FkDims d;
    AxisDesc ax_c1, ax_c2, ax_n1, ax_n2;
    const float2* twc;        // [C1 pos q][C2]   W_nx^{c2 * kc1(q)}
    const float2* twt;        // [N1 pos q1][N2]  W_M^{n2 * k1(q1)}
    const float2* win;        // [M] packed tukey(ns, 0.03): (w[2m], w[2m+1])
    const int* row_partner;   // [nx] row position of wavenumber -k
    const int* q1_partner;    // [N1] position of (N1 - k1) mod N1
    const int* mirror0;       // [N2] position of (N2 - k2) mod N2
    const float2* wrow;       // [N1] W_ns^{k1(q1)}
    const float2* wcol;       // [N2] W_ns^{N1 * k2(i)}
    const float* mask;        // [nx pos r][N1 pos q1][N2 pos i] folded mask M_h(k, f), f < M
    const float* nyq;         // [nx pos r] M_h(k, M)
    const int2* pairs;        // pass-B work list (keyA, keyB)
    float scale;              // 1 / (nx * M)
    // Bluestein form of the c2 sub-transform (C2 has a prime factor > 31): a length-C2 DFT as a circular
    // convolution of length bs_L = 2^a 3^b 5^c >= 2 C2 - 1 with the chirp exp(-i pi n^2 / C2); bs_L = 0: off
    int bs_L;
    AxisDesc ax_bs;           // the length-bs_L transform
    const float2* bs_chirp;   // [C2]    exp(-i pi n^2 / C2)
    const float2* bs_filt;    // [bs_L]  FFT of the conjugate chirp (wrapped) / bs_L, at the DIF positions of ax_bs
    // the same for the n2 sub-transform of pass B (ns / 2 has a prime factor > 31): bn_L = 0: off
    int bn_L;
    AxisDesc ax_bn;
    const float2* bn_chirp;   // [N2]
    const float2* bn_filt;    // [bn_L]
};

}  // namespace d4w

#include "fk_fast.h"
#include "fk_tf.h"

namespace d4w {

// ---------------------------------------------------------------------------------------------
// shape-specialised kernels (fk_fast.h): one table entry per instantiated shape
// ---------------------------------------------------------------------------------------------
struct FkFastEntry {
    int variant;           // D4W_FK_VARIANT picks among entries of one shape (0 = default)
    int nx, ns, C1, C2A, C2B, N1, NA, NB, NC, TA, TC, thrA, thrC, thrB;
    int C2X;               // > 1: the c2 axis (= C2X, with C2A = C2B = 1) runs the generic Bluestein pass C
    size_t ldsA, ldsC, ldsB;
    int wgA, wgC, wgB;     // resident workgroups per CU the persistent grids are sized for
    void (*A_fwd)(FkDev, const float2*, float2*, int, int, int, int, FkGeo);
    void (*A_fwd_taper)(FkDev, const float2*, float2*, int, int, int, int, FkGeo);
    void (*A_inv)(FkDev, float2*, int, int, int, int, FkGeo, const float2*);
    void (*A_inv_stats)(FkDev, float2*, int, int, double*, unsigned*, int, int, FkGeo, const float2*, int);
    void (*T_inv_stats)(FkDev, float2*, int, int, double*, unsigned*, int, int, FkGeo, const float2*, int);   // MODE 1
    void (*C_fwd)(FkDev, FkFastDev, float2*, int, int, int, int, FkGeo);
    void (*C_inv)(FkDev, FkFastDev, float2*, int, int, int, int, FkGeo);
    void (*B_mid)(FkDev, FkFastDev, float2*, int, int, FkGeo);
    // distributed (channel-sharded) layouts, fk_fast.h FkGeo
    void (*T_fwd)(FkDev, const float2*, float2*, int, int, int, int, FkGeo);          // pass A MODE 1 (time phase)
    void (*T_fwd_taper)(FkDev, const float2*, float2*, int, int, int, int, FkGeo);
    void (*T_inv)(FkDev, float2*, int, int, int, int, FkGeo, const float2*);
    void (*T_inv_env)(FkDev, float2*, int, int, int, int, FkGeo, const float2*, const float2*, int, const float*);   // + analytic-signal epilogue
    void (*Ac_fwd)(FkDev, const float2*, float2*, int, int, int, int, FkGeo);         // pass A MODE 2 (c1 transform on the slab)
    void (*Ac_inv)(FkDev, float2*, int, int, int, int, FkGeo, const float2*);
    void (*Cs_fwd)(FkDev, FkFastDev, float2*, int, int, int, int, FkGeo);             // pass C on the slab
    void (*Cs_inv)(FkDev, FkFastDev, float2*, int, int, int, int, FkGeo);
    void (*Bs_mid)(FkDev, FkFastDev, float2*, int, int, FkGeo);                       // pass B on the slab
    void (*Bs_hilb)(FkDev, FkFastDev, float2*, int, int, FkGeo);                      // ... with the Hilbert pair operation (real rows)
    // time-first order (fk_tf.h)
    size_t ldsBt;
    void (*Bt_fwd)(FkDev, FkFastDev, FkTfDev, float2*, int, int);                     // Bf: n2 forward + untangle -> W
    void (*Bt_inv)(FkDev, FkFastDev, FkTfDev, float2*, int, int);                     // Bi: W -> re-tangle + n2 inverse
    void (*C_mid)(FkDev, FkFastDev, FkTfDev, int, int);                               // Cm: c2 forward x mask x c2 inverse on W
};

template <class G>
static inline FkFastEntry fast_entry(int wgA, int wgC, int wgB, int variant = 0) {
    FkFastEntry e;
    e.variant = variant;
    e.nx = G::NX; e.ns = 2 * G::M;
    e.C1 = G::C1; e.C2A = G::C2A; e.C2B = G::C2B; e.N1 = G::N1; e.NA = G::NA; e.NB = G::NB; e.NC = G::NC;
    e.TA = G::TA; e.TC = G::TC; e.thrA = G::THRA; e.thrC = G::THRC; e.thrB = G::THRB;
    e.C2X = G::C2X;
    e.ldsA = G::ldsA; e.ldsC = G::ldsC; e.ldsB = G::ldsB;
    e.wgA = wgA; e.wgC = wgC; e.wgB = wgB;
    e.A_fwd = fkf_passA_fwd<G, false, 0>;
    e.A_fwd_taper = fkf_passA_fwd<G, true, 0>;
    e.A_inv = fkf_passA_inv<G, 0>;
    e.A_inv_stats = fkf_passA_inv_stats<G, 0>;
    e.T_inv_stats = fkf_passA_inv_stats<G, 1>;
    e.C_fwd = fkf_passC<G, false, 0>;
    e.C_inv = fkf_passC<G, true, 0>;
    e.B_mid = fkf_passB<G, 0>;
    e.T_fwd = fkf_passA_fwd<G, false, 1>;
    e.T_fwd_taper = fkf_passA_fwd<G, true, 1>;
    e.T_inv = fkf_passA_inv<G, 1>;
    e.T_inv_env = fkf_passA_inv_env<G>;
    e.Ac_fwd = fkf_passA_fwd<G, false, 2>;
    e.Ac_inv = fkf_passA_inv<G, 2>;
    e.Cs_fwd = fkf_passC<G, false, 1>;
    e.Cs_inv = fkf_passC<G, true, 1>;
    e.Bs_mid = fkf_passB<G, 1>;
    e.Bs_hilb = fkf_passB<G, 1, true>;
    e.ldsBt = G::ldsB + (size_t)G::NC * G::NB * sizeof(int2);
    e.Bt_fwd = fkf_passBt<G, 1>;
    e.Bt_inv = fkf_passBt<G, 2>;
    if constexpr (G::C2X == 1) e.C_mid = fkf_passCm<G>;
    else e.C_mid = nullptr;                 // the c2 axis runs as a Bluestein convolution: fk_passCm_bluestein (fk_filter.hip)
    return e;
}

}  // namespace d4w
