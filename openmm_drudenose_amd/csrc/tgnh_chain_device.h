// tgnh_chain_device.h -- device code of the Nose-Hoover chain (A5), shared by chain_kernel and by the tile
// kernel's in-kernel chain.  Included by tgnh_kernels.hip only.
//
// TGNH: one lane per thermostat (Cu :558-650).  dualNH: one lane runs the reference's coupled, interleaved
// vectors (Ref :467-504), including its indexing quirk when useDrudeNHChains is false.
// The chain variables are copied into registers (numNHChains <= 4, fully unrolled) or LDS (longer chains) for
// the S-fold loop and written back once.
#ifndef TGNH_CHAIN_DEVICE_H_
#define TGNH_CHAIN_DEVICE_H_

namespace tgnh {
#ifdef TGNH_TRACE
static __device__ __attribute__((unused)) double g_chain_dbg[4];   // largest exponent argument seen by the real / Drude fast paths, times each left the polynomial's range
#endif

// Contraction is left on: every a*b+c below may become one fma (<= 1 ulp per operation away from the separately
// rounded reference arithmetic; the chain is smooth, the parity gate is 1e-6 and is met at 1e-12).
#pragma clang fp contract(fast)

// exp() for the chain.  The arguments are -dtc/8*etaDot and -dtc/2*etaDot: exactly 0 for the dummy link
// (exp(-0) = 1 exactly, as libm returns) and tiny otherwise, so a short Taylor polynomial in explicit FMAs is
// exact to double rounding: degree 6 for |x| < 2^-7 (truncation |x|^7/7! < 2^-61 relative), degree 11 for
// |x| < 2^-5 (< 2^-55); larger arguments take the library exp.  The chain is one serial fp64 chain per thermostat
// on the critical path of every time step, and a wavefront issues one fp64 instruction per 4 cycles however few
// lanes are live, so what counts is the *number of instructions*: Horner (an Estrin form, 14 operations at depth 5,
// measured 40 % slower), and the degree is chosen for the whole wavefront so that thermostats on different sides of
// the switch do not execute both polynomials.
// exp(x) for |x| < 1 in 17 instructions and no branch: the degree-14 Taylor polynomial of x/4 (truncation (1/4)^15/15! < 2^-70),
// squared twice (<= 5 ulp).  A thermostat far from equilibrium -- Drude particles much hotter than their 1 K, as in the
// synthetic boxes of bench.py, where -dtc/2*etaDot reaches 0.3 -- leaves the short polynomials' range at every sub-step.
__device__ __forceinline__ double chain_exp_wide(const double x) {
    const double y = 0.25 * x;
    double p = 1.0 / 87178291200.0;
    p = fma(p, y, 1.0 / 6227020800.0);
    p = fma(p, y, 1.0 / 479001600.0);
    p = fma(p, y, 1.0 / 39916800.0);
    p = fma(p, y, 1.0 / 3628800.0);
    p = fma(p, y, 1.0 / 362880.0);
    p = fma(p, y, 1.0 / 40320.0);
    p = fma(p, y, 1.0 / 5040.0);
    p = fma(p, y, 1.0 / 720.0);
    p = fma(p, y, 1.0 / 120.0);
    p = fma(p, y, 1.0 / 24.0);
    p = fma(p, y, 1.0 / 6.0);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.0);
    p = fma(p, y, 1.0);
    p *= p;
    return p * p;
}

template <bool LIBM = true>
__device__ __forceinline__ double chain_exp(double x) {
    const double ax = fabs(x);
    if (__builtin_expect(!__any(ax >= 0.0078125), 1)) {              // also x = +-0: returns exactly 1
        double p = 1.0 / 720.0;
        p = fma(p, x, 1.0 / 120.0);
        p = fma(p, x, 1.0 / 24.0);
        p = fma(p, x, 1.0 / 6.0);
        p = fma(p, x, 0.5);
        p = fma(p, x, 1.0);
        return fma(p, x, 1.0);
    }
    if (LIBM ? ax < 0.03125 : !__any(ax >= 0.03125)) {
        double p = 1.0 / 39916800.0;
        p = fma(p, x, 1.0 / 3628800.0);
        p = fma(p, x, 1.0 / 362880.0);
        p = fma(p, x, 1.0 / 40320.0);
        p = fma(p, x, 1.0 / 5040.0);
        p = fma(p, x, 1.0 / 720.0);
        p = fma(p, x, 1.0 / 120.0);
        p = fma(p, x, 1.0 / 24.0);
        p = fma(p, x, 1.0 / 6.0);
        p = fma(p, x, 0.5);
        p = fma(p, x, 1.0);
        return fma(p, x, 1.0);
    }
    if (LIBM) return exp(x);
    if (!__any(ax >= 1.0)) return chain_exp_wide(x);
    // no libm call (the streaming kernels' in-kernel chains: ocml's exp would cost them ~30 VGPRs): x = n ln 2 + r with
    // |r| <= 0.35, exp(r) from chain_exp_wide, the power of two by ldexp; <= 6 ulp for every argument exp() itself can take
    x = fmax(-750.0, fmin(750.0, x));                                // (exp(+-inf) = inf, 0 as libm's: a chain that has overflowed goes on as the reference's does)
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -0.693147180369123816490, x);                  // ln 2 in two pieces (the first with 21 trailing zero bits: n * hi is exact)
    r = fma(n, -1.90821492927058770002e-10, r);
    return ldexp(chain_exp_wide(r), (int)n);
}

// a * b + c as exactly one v_fma_f64 on three vector registers
__device__ __forceinline__ double fma3(const double a, const double b, const double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// exp(y) for |y| < 2^-4 in nine fused multiply-adds (Horner, degree 9: truncation y^10 / 10! < 2^-61): the instruction count of the depth-3 Estrin form of
// degree 7 that stood here until round 4 (|y| < 2^-6), with four times its range.  A lone wavefront issues one fp64 instruction per ~8 cycles, 9.5 when it depends on the one before
// (tools/micro/issue_probe.hip): the depth Estrin's form saves is worth 13 cycles an exponential, the eight instructions of the
// wide form that a box between 2^-6 and 2^-4 no longer needs are worth 64 -- and the synthetic boxes of bench.py sit there
// (arguments 0.02-0.03 at 32 k slots with three or ten links).
__device__ __forceinline__ double chain_exp9(const double y) {
    double p = 1.0 / 362880.0;
    p = fma(p, y, 1.0 / 40320.0);
    p = fma(p, y, 1.0 / 5040.0);
    p = fma(p, y, 1.0 / 720.0);
    p = fma(p, y, 1.0 / 120.0);
    p = fma(p, y, 1.0 / 24.0);
    p = fma(p, y, 1.0 / 6.0);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.0);
    return fma(p, y, 1.0);
}

struct ChainConst {
    double dtc2, dtc4, dtc8;
    int S;
};

// The S sub-steps of a register-resident chain of CC >= 2 links with the exponentials that repeat taken once (see
// chain_real_core) and no range test inside the loop: WIDE = false evaluates them with the degree-9 polynomial (|x| < 2^-4),
// WIDE = true with chain_exp_wide (|x| < 1).  Returns the largest |x| met; the caller repeats the call with the next wider
// form if that left the range.  `live`: the etaMass > 0 guard of link 0 (Cu :561, :579; the Drude thermostat has none).
template <int CC, bool WIDE>
__device__ __forceinline__ double chain_fast_loop(double* eta, double* etaDot, double* etaDotDot, const double* etaMass,
                                                  const double* invM, const ChainConst k, const double nkbt, const double kbT,
                                                  const bool live, const double invQ0, const double ef_top, double& ke, double& scale) {
    double ef[CC], xmax = 0.0;
    ef[CC - 1] = ef_top;                                             // the dummy link never moves: constant of the call
    for (int iter = 0; iter < k.S; iter++) {
#pragma unroll
        for (int i = CC - 1; i >= 0; i--) {                          // Cu :566-571 / :607-618
            if (i < CC - 1) { const double x = -k.dtc8 * etaDot[i + 1]; xmax = fmax(xmax, fabs(x)); ef[i] = WIDE ? chain_exp_wide(x) : chain_exp9(x); }
            etaDot[i] *= ef[i];
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= ef[i];
        }
        { const double x = -k.dtc2 * etaDot[0]; xmax = fmax(xmax, fabs(x)); const double e = WIDE ? chain_exp_wide(x) : chain_exp9(x); scale *= e; ke *= e * e; }   // Cu :573-574 / :620-621
#pragma unroll
        for (int i = 0; i < CC; i++) eta[i] += k.dtc2 * etaDot[i];   // Cu :575-577 / :623
        if (live) etaDotDot[0] = (ke - nkbt) * invQ0;                // Cu :579-581 / :629
        etaDot[0] *= ef[0];                                          // Cu :583-585 / :630-632
        etaDot[0] += etaDotDot[0] * k.dtc4;
        etaDot[0] *= ef[0];
#pragma unroll
        for (int i = 1; i < CC; i++) {                               // Cu :586-592 / :633-641, expfac as in the descending loop
            etaDot[i] *= ef[i];
            etaDotDot[i] = (etaMass[i - 1] * etaDot[i - 1] * etaDot[i - 1] - kbT) * invM[i];
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= ef[i];
        }
    }
    return xmax;
}

// Fast forms first (chain_fast_loop), the transcription last: true when one of the fast forms held its range and the chain
// is done.  The form is chosen for the whole wavefront from the exponents at entry, with a factor two of room.
template <int CC, bool LIBM>
__device__ __forceinline__ bool chain_fast(double* eta, double* etaDot, double* etaDotDot, const double* etaMass, const double* invM,
                                           const ChainConst k, const double nkbt, const double kbT, const bool live,
                                           const double invQ0, double& ke, double& scale, const int which) {
    double s_eta[CC], s_ed[CC + 1], s_edd[CC];
#pragma unroll
    for (int i = 0; i < CC; i++) { s_eta[i] = eta[i]; s_ed[i] = etaDot[i]; s_edd[i] = etaDotDot[i]; }
    s_ed[CC] = etaDot[CC];
    const double ke_in = ke;
    const double ef_top = chain_exp<LIBM>(-k.dtc8 * etaDot[CC]);
    double x0 = fabs(k.dtc2 * etaDot[0]);
#pragma unroll
    for (int i = 1; i < CC; i++) x0 = fmax(x0, fabs(k.dtc8 * etaDot[i]));
    bool wide = __any(x0 >= 0.03125);
    double xmax = 0.0;
    if (!wide) {
        xmax = chain_fast_loop<CC, false>(eta, etaDot, etaDotDot, etaMass, invM, k, nkbt, kbT, live, invQ0, ef_top, ke, scale);
#ifdef TGNH_TRACE
        if (blockIdx.x == 0) { g_chain_dbg[which] = fmax(g_chain_dbg[which], xmax); if (xmax >= 0.0625) g_chain_dbg[2 + which] += 1.0; }
#endif
        if (__builtin_expect(!__any(xmax >= 0.0625), 1)) return true;
        wide = true;
    } else {
        xmax = 1.0;                                                  // (nothing run yet: fall into the wide form)
    }
    // restore and run the wide form
#pragma unroll
    for (int i = 0; i < CC; i++) { eta[i] = s_eta[i]; etaDot[i] = s_ed[i]; etaDotDot[i] = s_edd[i]; }
    etaDot[CC] = s_ed[CC];
    ke = ke_in; scale = 1.0;
    if (live) etaDotDot[0] = (ke - nkbt) * invQ0;
    xmax = chain_fast_loop<CC, true>(eta, etaDot, etaDotDot, etaMass, invM, k, nkbt, kbT, live, invQ0, ef_top, ke, scale);
#ifdef TGNH_TRACE
    if (blockIdx.x == 0) { g_chain_dbg[which] = fmax(g_chain_dbg[which], xmax); if (xmax >= 1.0) g_chain_dbg[2 + which] += 1000.0; }
#endif
    if (__builtin_expect(!__any(xmax >= 1.0), 1)) return true;
#pragma unroll
    for (int i = 0; i < CC; i++) { eta[i] = s_eta[i]; etaDot[i] = s_ed[i]; etaDotDot[i] = s_edd[i]; }
    etaDot[CC] = s_ed[CC];
    ke = ke_in; scale = 1.0;
    if (live) etaDotDot[0] = (ke - nkbt) * invQ0;
    return false;
}

// One real (temperature-group or COM) thermostat.  Cu :560-595.  CC > 0: compile-time chain length.
template <int CC, bool LIBM = true, bool TRY_FAST = true>
__device__ __forceinline__ void chain_real_core(double* eta, double* etaDot, double* etaDotDot, const double* etaMass,
                                                const int Cdyn, const ChainConst k, const double nkbt, const double kbT,
                                                double ke, double* scale_out, double* ke_out) {
    const int C = CC > 0 ? CC : Cdyn;
    double scale = 1.0, expfac = 1.0;
    const bool live = etaMass[0] > 0;
    const double invQ0 = live ? 1.0 / etaMass[0] : 0.0;              // (KE - NkT)/Q as a multiply: <= 1 ulp from the division
    if (live) etaDotDot[0] = (ke - nkbt) * invQ0;                    // Cu :561-563
    if constexpr (CC == 1) {
        // One link: link 1 is the reference's dummy that "will always have etaDot = 0" (Cu :252, Ref :215), so
        // expfac = exp(-dtc8*0) = 1 exactly and each of Cu :568-570 / :583-585 collapses to one fused multiply-add.
        // This is the serial critical path of a whole time step: 16 dependent fp64 operations per sub-step.
        double ed = etaDot[0], edd = etaDotDot[0], et = eta[0];
        for (int iter = 0; iter < k.S; iter++) {
            ed = fma(edd, k.dtc4, ed);                               // Cu :568-570 with expfac = 1
            const double e = chain_exp<LIBM>(-k.dtc2 * ed);
            scale *= e; ke *= e * e;                                 // Cu :573-574
            et = fma(k.dtc2, ed, et);                                // Cu :575-577
            if (live) edd = (ke - nkbt) * invQ0;                     // Cu :579-581
            ed = fma(edd, k.dtc4, ed);                               // Cu :583-585
        }
        etaDot[0] = ed; etaDotDot[0] = edd; eta[0] = et;
        *scale_out = scale;
        *ke_out = ke;
        return;
    }
    constexpr int CMI = CC > 0 ? CC : 1;
    double invM[CMI];                                                // 1/Q of the higher links (CC == 0: divide)
#pragma unroll
    for (int i = 0; i < CMI; i++) invM[i] = (CC > 0) ? 1.0 / etaMass[i] : 0.0;
    if constexpr (CC >= 2 && TRY_FAST) {
        // Register-resident chains: the same arithmetic with the exponentials that repeat taken once.  Per sub-step
        // the reference evaluates expfac = exp(-dtc8*etaDot[i+1]) in the descending loop (Cu :566-571) and AGAIN in
        // the ascending loop (Cu :586-592), where etaDot[i+1] still holds the descending loop's value: same argument,
        // same result, computed once (C instead of 2C exponentials per sub-step, none for the top link, whose
        // neighbour is the constant dummy).  exp() is a polynomial without a range test (chain_fast_loop); the largest
        // argument seen is checked once after the loop and the call repeated with the next wider form, last of all with
        // the careful code below (|x| >= 1: never in practice).
        if (chain_fast<CC, LIBM>(eta, etaDot, etaDotDot, etaMass, invM, k, nkbt, kbT, live, invQ0, ke, scale, 0)) {
            *scale_out = scale;
            *ke_out = ke;
            return;
        }
    }
    for (int iter = 0; iter < k.S; iter++) {
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {                           // Cu :566-571
            expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[i + 1]);
            etaDot[i] *= expfac;
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= expfac;
        }
        { const double e = chain_exp<LIBM>(-k.dtc2 * etaDot[0]); scale *= e; ke *= e * e; }   // Cu :573-574, exp(-dtc x) = e^2
#pragma unroll
        for (int i = 0; i < C; i++) eta[i] += k.dtc2 * etaDot[i];    // Cu :575-577
        if (live) etaDotDot[0] = (ke - nkbt) * invQ0;                // Cu :579-581
        etaDot[0] *= expfac;                                         // Cu :583-585 (expfac of link 0 reused)
        etaDot[0] += etaDotDot[0] * k.dtc4;
        etaDot[0] *= expfac;
#pragma unroll
        for (int i = 1; i < C; i++) {                                // Cu :586-592
            expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[i + 1]);
            etaDot[i] *= expfac;
            etaDotDot[i] = (CC > 0) ? (etaMass[i - 1] * etaDot[i - 1] * etaDot[i - 1] - kbT) * invM[i < CMI ? i : 0]
                                    : (etaMass[i - 1] * etaDot[i - 1] * etaDot[i - 1] - kbT) / etaMass[i];
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= expfac;
        }
    }
    *scale_out = scale;
    *ke_out = ke;
}

// The Drude thermostat.  Cu :597-642.
template <int CC, bool LIBM = true, bool TRY_FAST = true>
__device__ __forceinline__ void chain_drude_core(double* eta, double* etaDot, double* etaDotDot, const double* etaMass,
                                                 const int Cdyn, const bool chains, const ChainConst k, const double nkbt,
                                                 const double kbT, double ke, double* scale_out, double* ke_out) {
    const int C = CC > 0 ? CC : Cdyn;
    double scale = 1.0, expfac = 1.0;
    const double invQ0 = 1.0 / etaMass[0];
    etaDotDot[0] = (ke - nkbt) * invQ0;                              // Cu :605
    if (CC == 1 || !chains) {
        // Only link 0 moves (Cu :607-614, :624-628, :633-641 are skipped), so expfac = exp(-dtc8*etaDot[1]) is a
        // constant of the whole call (1 exactly when etaDot[1] = 0, the usual case).
        expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[1]);                     // Cu :615
        const double ef2 = expfac * expfac, efd = expfac * k.dtc4;
        double ed = etaDot[0], edd = etaDotDot[0], et = eta[0];
        const bool unit = expfac == 1.0;
        for (int iter = 0; iter < k.S; iter++) {
            ed = unit ? fma(edd, k.dtc4, ed) : fma(ed, ef2, edd * efd);      // Cu :616-618: (ed*ef + edd*dtc4)*ef
            const double e = chain_exp<LIBM>(-k.dtc2 * ed);
            scale *= e; ke *= e * e;                                 // Cu :620-621
            et = fma(k.dtc2, ed, et);                                // Cu :623
            edd = (ke - nkbt) * invQ0;                               // Cu :629
            ed = unit ? fma(edd, k.dtc4, ed) : fma(ed, ef2, edd * efd);      // Cu :630-632
        }
        etaDot[0] = ed; etaDotDot[0] = edd; eta[0] = et;
        *scale_out = scale;
        *ke_out = ke;
        return;
    }
    if constexpr (CC >= 2 && TRY_FAST) {                             // as in chain_real_core: repeated exponentials once
        double invM[CC];
#pragma unroll
        for (int i = 0; i < CC; i++) invM[i] = 1.0 / etaMass[i];
        if (chain_fast<CC, LIBM>(eta, etaDot, etaDotDot, etaMass, invM, k, nkbt, kbT, true, invQ0, ke, scale, 1)) {
            *scale_out = scale;
            *ke_out = ke;
            return;
        }
    }
    for (int iter = 0; iter < k.S; iter++) {                         // Cu :606-642
#pragma unroll
        for (int i = C - 1; i > 0; i--) {
            expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[i + 1]);
            etaDot[i] *= expfac;
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= expfac;
        }
        expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[1]);
        etaDot[0] *= expfac;
        etaDot[0] += etaDotDot[0] * k.dtc4;
        etaDot[0] *= expfac;
        { const double e = chain_exp<LIBM>(-k.dtc2 * etaDot[0]); scale *= e; ke *= e * e; }   // Cu :620-621
        eta[0] += k.dtc2 * etaDot[0];
#pragma unroll
        for (int i = 1; i < C; i++) eta[i] += k.dtc2 * etaDot[i];
        etaDotDot[0] = (ke - nkbt) * invQ0;
        etaDot[0] *= expfac;
        etaDot[0] += etaDotDot[0] * k.dtc4;
        etaDot[0] *= expfac;
#pragma unroll
        for (int i = 1; i < C; i++) {
            expfac = chain_exp<LIBM>(-k.dtc8 * etaDot[i + 1]);
            etaDot[i] *= expfac;
            etaDotDot[i] = (etaMass[i - 1] * etaDot[i - 1] * etaDot[i - 1] - kbT) / etaMass[i];
            etaDot[i] += etaDotDot[i] * k.dtc4;
            etaDot[i] *= expfac;
        }
    }
    *scale_out = scale;
    *ke_out = ke;
}

// The fast forms for real thermostats AND the Drude thermostat (with its own chain: useDrudeNHChains) in ONE call, converged: what
// chain_real_core and chain_drude_core do before and inside chain_fast differs in parameters only -- kT, NkT, the guard of link 0
// (Cu :561 / :579 have it, :605 / :629 do not) -- and those ride in the lane.  Called thermostat by thermostat the two kinds
// are two code paths of a wavefront, one AFTER the other: where one wavefront holds them all (chainN_run: the chains of 2-4
// links inside the streaming launches) that doubled the serial section every work-group of the launch waits for.  The form
// (narrow / wide polynomial) is chosen for all lanes of the call together, as before for all lanes of a kind.
// False: the arguments left every fast form's range (|x| >= 1: never in practice); the state is as it was.
template <int CC, bool LIBM>
__device__ __forceinline__ bool chain_both_fast(double* eta, double* etaDot, double* etaDotDot, const double* etaMass, const ChainConst k,
                                                const double nkbt, const double kbT, const bool drude, const double ke_in,
                                                double* scale_out, double* ke_out) {
    const bool live = drude || etaMass[0] > 0;
    const double invQ0 = live ? 1.0 / etaMass[0] : 0.0;
    const double edd0 = etaDotDot[0];
    if (live) etaDotDot[0] = (ke_in - nkbt) * invQ0;                 // Cu :561-563 / :605
    double invM[CC];
#pragma unroll
    for (int i = 0; i < CC; i++) invM[i] = 1.0 / etaMass[i];
    double ke = ke_in, scale = 1.0;
    if (chain_fast<CC, LIBM>(eta, etaDot, etaDotDot, etaMass, invM, k, nkbt, kbT, live, invQ0, ke, scale, drude ? 1 : 0)) {
        *scale_out = scale;
        *ke_out = ke;
        return true;
    }
    etaDotDot[0] = edd0;                                             // (the transcription sets it again: nothing else has changed)
    return false;
}

// One TGNH thermostat (lane itg), `reps` chain calls back to back on register / LDS copies.  BOTH: the calling wavefront holds real
// thermostats and the Drude thermostat (chainN_run) -- all of them through chain_both_fast in one pass; the chain kernels keep the Drude
// thermostat on a wavefront of its own and the two cores as they were (the merged form costs chain_long_kernel<10> 66 more registers,
// spilled to accumulation registers, and 3 us of its 36).
template <int CC, bool LIBM = true, bool BOTH = false>
// st_in -> st_out (may alias); `write`: this caller owns the write-back; s_scale (LDS, may be null) receives the factors.
__device__ __forceinline__ void run_tgnh(const ChainArgs& a, const double* st_in, double* st_out, const bool write,
                                         double* s_scale, const int itg, double* lds, const double ke_in) {
    const ChainLayout& L = a.L;
    const int C = L.C, NT = L.NT;
    constexpr int CM = CC > 0 ? CC : 1;
    double r_eta[CM], r_etaDot[CM + 1], r_etaDotDot[CM], r_etaMass[CM];
    double *eta = r_eta, *etaDot = r_etaDot, *etaDotDot = r_etaDotDot, *etaMass = r_etaMass;
    if (CC == 0) {                       // long chains: this lane's slice of the LDS scratch
        eta = lds + itg * (4 * C + 1); etaDot = eta + C; etaDotDot = etaDot + C + 1; etaMass = etaDotDot + C;
    }
    const double* g_eta = st_in + L.off_eta + itg * C;
    const double* g_etaDot = st_in + L.off_etaDot + itg * (C + 1);
    const double* g_etaDotDot = st_in + L.off_etaDotDot + itg * C;
    const double* g_etaMass = st_in + L.off_etaMass + itg * C;
#pragma unroll
    for (int i = 0; i < (CC > 0 ? CC : C); i++) { eta[i] = g_eta[i]; etaDotDot[i] = g_etaDotDot[i]; etaMass[i] = g_etaMass[i]; }
#pragma unroll
    for (int i = 0; i < (CC > 0 ? CC : C) + 1; i++) etaDot[i] = g_etaDot[i];
    ChainConst k;
    const double dtc = a.dt / a.S;                                   // Cu :440-443
    k.dtc2 = dtc / 2.0; k.dtc4 = dtc / 4.0; k.dtc8 = dtc / 8.0; k.S = a.S;
    const double nkbt = st_in[L.off_nkbt + itg];
    double ke = ke_in;
    if (write) st_out[L.off_ke + itg] = ke;                          // KE before the chain (Cu :490)
    const int reps = a.chain_twice ? 2 : 1;
    double total = 1.0;
    for (int rep = 0; rep < reps; rep++) {
        double sc = 1.0, kep = ke;
        const bool drude = itg == NT - 1;
        bool done = false;
        if constexpr (CC >= 2 && BOTH) {                             // every lane whose chain has CC moving links, in one call (chain_both_fast)
            if (!drude || L.use_drude_chains != 0)
                done = chain_both_fast<CC, LIBM>(eta, etaDot, etaDotDot, etaMass, k, nkbt, drude ? a.drudekbT : a.realkbT, drude, ke, &sc, &kep);
        }
        if (!done) {
            if (!drude) chain_real_core<CC, LIBM, !BOTH>(eta, etaDot, etaDotDot, etaMass, C, k, nkbt, a.realkbT, ke, &sc, &kep);
            else chain_drude_core<CC, LIBM, !BOTH>(eta, etaDot, etaDotDot, etaMass, C, L.use_drude_chains != 0, k, nkbt, a.drudekbT, ke, &sc, &kep);
        }
        if (write) {
            if (rep == 0) { st_out[L.off_scale_a + itg] = sc; st_out[L.off_ke_post + itg] = kep; }
            else st_out[L.off_scale_b + itg] = sc;
        }
        total *= sc;
        ke = kep;
    }
    if (s_scale) s_scale[itg] = total;
    if (!write) return;
    if (reps == 1) st_out[L.off_scale_b + itg] = 1.0;
    st_out[L.off_scale + itg] = total;
    double* o_eta = st_out + L.off_eta + itg * C;
    double* o_etaDot = st_out + L.off_etaDot + itg * (C + 1);
    double* o_etaDotDot = st_out + L.off_etaDotDot + itg * C;
#pragma unroll
    for (int i = 0; i < (CC > 0 ? CC : C); i++) { o_eta[i] = eta[i]; o_etaDotDot[i] = etaDotDot[i]; }
#pragma unroll
    for (int i = 0; i < (CC > 0 ? CC : C) + 1; i++) o_etaDot[i] = etaDot[i];
}

// One-link TGNH thermostat in two halves, for the in-kernel chain of the rescale launches: the state loads are issued
// at kernel entry, *ahead of* the tile loads (behind them they would queue for microseconds while every work-group
// waits on its scale factors), the arithmetic runs while the tile loads are in flight.  Same arithmetic as
// run_tgnh<1, false>.
struct Chain1Regs { double eta, etaDot0, etaDot1, etaDotDot, etaMass, nkbt, ke; };
// Where thermostat-lane itg (index into the NT-long KE / scale vectors) keeps its one-link chain in the block.
//   TGNH:   [thermostat][link] rows, etaDot rows of 2 (link 0, dummy).
//   dualNH with useDrudeNHChains (the self-consistent layout, Ref :186-217): the interleaved vectors
//           [real0, drude0 | dummy, dummy]; lanes 0 (real) and 2 (Drude) of the internal [real, unused, Drude] order.
//           Link i couples to link i+2 there (Ref :477, :495), i.e. to its dummy: two independent one-link chains --
//           the arithmetic of the TGNH ones (SURVEY A9's bridge identity), so they take the same code.
struct Chain1Map { int eta, ed0, ed1, edd, mass; bool used, guard; };
__device__ __forceinline__ Chain1Map chain1_map(const ChainLayout& L, const int itg) {
    Chain1Map m;                                                  // branch-free: the coefficients are the layout's
    const int t = itg >> L.c1_shift;
    m.eta = t; m.edd = t; m.mass = t;
    m.ed0 = t * L.c1_mul; m.ed1 = m.ed0 + L.c1_add;
    m.used = itg != L.c1_unused;
    m.guard = itg < L.c1_guard_below;                             // etaMass > 0 guard: TGNH's real thermostats (Cu :561 vs :605, Ref :471)
    return m;
}

// ---- mailbox exchange (protocol: XchgArgs in tgnh_internal.h) ----
__device__ __forceinline__ size_t xchg_cell(const XchgArgs& x, const unsigned par, const int src, const int i, const int copy = 0) {
    return (size_t)copy * XCHG_REPLICA_U64 + (((size_t)par * x.world + src) * XCHG_NT_PAD + i) * XCHG_CELL_U64;
}
__device__ __forceinline__ unsigned long long xchg_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The mailbox thread tid of a sending work-group stores into: thread tid serves peer tid % world, so every thread needs ONE
// mailbox pointer, which a caller with time to spare fetches ahead of the send (step_kernel: before it collects the rows).
__device__ __forceinline__ unsigned long long* xchg_peer_of(const XchgArgs& x, const int tid) {
    return x.world == 1 ? x.mine : x.peers[tid % x.world];
}
// Called by one work-group with `mine` = this rank's sum in thread tid < NT, handed over through s_val (LDS, NT doubles).
// Contains __syncthreads().  seq_new != 0 (thread 0): the number of this exchange, when the caller has read the counter
// already (step_kernel reads it at kernel entry: no load on the path between the last row and the send).
// peer = xchg_peer_of(x, tid) when the caller has fetched it already.
__device__ __forceinline__ void xchg_send(const XchgArgs& x, const int NT, const int tid, const int nthreads, double* s_val,
                                          const double mine, const unsigned long long seq_new = 0ull,
                                          unsigned long long* peer = nullptr) {
    __shared__ unsigned long long s_seq;
    if (!peer) peer = xchg_peer_of(x, tid);
    if (tid == 0) { const unsigned long long s = seq_new ? seq_new : *x.seq + 1ull; *x.seq = s; s_seq = s; }
    if (tid < NT) s_val[tid] = mine;
    __syncthreads();
    const unsigned long long seq = s_seq, tag = (seq & 0xffffffffull) << 32;
    const int tpp = nthreads / x.world;                      // threads per peer
    if (tid >= tpp * x.world) return;
    // every copy of the peer's cells [parity][my rank][0 .. NT): copy-major, so copy 0 goes out first
    unsigned long long* const base = peer + xchg_cell(x, (unsigned)(seq & 1ull), x.rank, 0, 0);
    for (int q = tid / x.world; q < NT * XCHG_REPLICAS; q += tpp) {
        const int copy = q / NT, i = q - copy * NT;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s_val[i]);
        unsigned long long* cell = base + (size_t)copy * XCHG_REPLICA_U64 + (size_t)i * XCHG_CELL_U64;
        __hip_atomic_store(cell, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(cell + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Called by all 64 lanes of one wavefront, converged; s_val = LDS scratch of world*NT doubles owned by that wavefront.
// Returns the all-rank sum of thermostat `lane` (lanes < NT).  seq_expected != 0: the exchange to wait for when this
// rank's own send may not have happened yet (step_kernel: sender and waiters are work-groups of one launch).
// SPREAD: the work-groups of the launch poll different copies of the cells (step_kernel, where all of them wait at the
// same moment); otherwise copy 0.
template <bool SPREAD = false>
__device__ __forceinline__ double xchg_wait_sum(const XchgArgs& x, const int NT, const int lane, double* s_val,
                                                const unsigned long long seq_expected = 0ull, bool* failed = nullptr) {
    const int cells = x.world * NT;
    const bool stamp = x.stat != nullptr && blockIdx.x == 0;            // work-group 0 keeps the rank's wait statistics
    const unsigned long long t_in = stamp ? wall_clock64() : 0ull;
    const unsigned long long* const box = x.mine + (SPREAD ? (size_t)(blockIdx.x % (unsigned)XCHG_REPLICAS) * XCHG_REPLICA_U64 : 0);
    // first batch: counter, latch and both parities of this lane's first cell, all in flight together
    const unsigned long long seq_raw = seq_expected ? seq_expected : __hip_atomic_load(x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned dead = __hip_atomic_load(x.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (lane < cells) {
        const int r = lane / NT, i = lane - r * NT;
        const unsigned long long* c0 = box + xchg_cell(x, 0u, r, i);
        const unsigned long long* c1 = box + xchg_cell(x, 1u, r, i);
        a0 = xchg_ld(c0); a1 = xchg_ld(c0 + 1); b0 = xchg_ld(c1); b1 = xchg_ld(c1 + 1);
    }
    const unsigned par = (unsigned)(seq_raw & 1ull);
    const unsigned long long tag = seq_raw & 0xffffffffull;
    bool timed_out = false;
    for (int k = lane; k < cells; k += 64) {
        const int r = k / NT, i = k - r * NT;
        const unsigned long long* c = box + xchg_cell(x, par, r, i);
        unsigned long long w0, w1;
        if (k == lane) { w0 = par ? b0 : a0; w1 = par ? b1 : a1; }
        else { w0 = xchg_ld(c); w1 = xchg_ld(c + 1); }
        unsigned n = 0;
        while (((w0 >> 32) != tag || (w1 >> 32) != tag) && dead == 0u && !timed_out) {
            if (++n > XCHG_SPIN_LIMIT) { timed_out = true; break; }
            // the latch may be set while this wavefront is already polling (step_kernel's work-group 0 giving up on a row, another
            // wavefront's time-out): looked at again every 64 polls, so that a failure ends every wait within microseconds
            if ((n & 63u) == 0u) dead = __hip_atomic_load(x.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(4);
            w0 = xchg_ld(c); w1 = xchg_ld(c + 1);
        }
        s_val[k] = __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
    }
    if (timed_out) {
        atomicOr(x.status, 4u);
        __hip_atomic_store(x.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (failed) *failed = __any(timed_out || dead != 0u);
    if (stamp) {                                                        // from "my sums are out" to "everybody's are here"
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long dt = wall_clock64() - t_in;
        if (lane == 0) {                                                // (one writer per launch: plain read-modify-write)
            x.stat[0] += dt;
            if (dt > x.stat[1]) x.stat[1] = dt;
            x.stat[2] += 1ull;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s = 0.0;
    if (lane < NT)
        for (int r = 0; r < x.world; r++) s += s_val[r * NT + lane];       // rank order, on every rank
    return s;
}

// Fixed-order sum of a FEW partial rows by one wavefront (every work-group computes the same bits).  The rows are
// read as one flat array: with W = the largest multiple of NT <= 64, lane l < W adds the elements l, l+W, l+2W, ...
// -- all of column l % NT -- into a single accumulator (independent loads), likewise the big-molecule rows at
// GRID_CAP; then one 64-lane butterfly per thermostat over the lanes of its column.  Returns this lane's thermostat
// sum (lane itg) and sum_b KE_b in thermostat order.
__device__ __forceinline__ void chain_sum_rows(const ChainArgs& a, const int lane, double* mine, double* kesum) {
    const int NT = a.L.NT;
    const int W = 64 - 64 % NT;
    const int n = a.nparts * NT, nb = a.nbig * NT;
    const double* big = a.partials + (size_t)GRID_CAP * NT;
    double acc = 0.0;
    if (lane < W) {
#pragma unroll 8
        for (int f = lane; f < n; f += W) acc += a.partials[f];
        for (int f = lane; f < nb; f += W) acc += big[f];
    }
    const int col = lane < W ? lane % NT : -1;
    double m = 0.0, s = 0.0;
    for (int b = 0; b < NT; b++) {
        const double t = wave_sum(col == b ? acc : 0.0);
        m = lane == b ? t : m;
        s += t;
    }
    *mine = m; *kesum = s;
}

// Where a chain that sums nothing itself finds its kinetic energies: the summed (and all-reduced) bins, or -- carried over,
// TGNH_FLAG_TRUST_STATE_CHANGED -- what the last chain left as KE * prod exp(-dtc etaDot) (Cu :574): every bin of the velocities
// its rescale produced is exactly s^2 times the bin before, so no pass over the velocities is needed to know them.
__device__ __forceinline__ int chain_ke_src(const ChainArgs& a) { return a.ke_carry ? a.L.off_ke_post : a.L.off_ke_red; }

__device__ __forceinline__ Chain1Regs chain1_load(const ChainArgs& a, const double* st_in, const int itg) {
    const ChainLayout& L = a.L;
    const Chain1Map m = chain1_map(L, itg);
    Chain1Regs r{};
    r.ke = st_in[chain_ke_src(a) + itg];
    if (!m.used) { r.etaMass = 1.0; return r; }
    r.eta = st_in[L.off_eta + m.eta];
    r.etaDot0 = st_in[L.off_etaDot + m.ed0];
    r.etaDot1 = st_in[L.off_etaDot + m.ed1];
    r.etaDotDot = st_in[L.off_etaDotDot + m.edd];
    r.etaMass = st_in[L.off_etaMass + m.mass];
    r.nkbt = st_in[L.off_nkbt + itg];
    return r;
}

// All NT thermostats in the lanes of ONE wavefront, one instruction stream: with one link the real thermostats
// (chain_real_core<1>) and the Drude thermostat (chain_drude_core<1>) differ only in expfac = exp(-dtc8*etaDot[1]),
// which is exactly 1 for the real ones (their link 1 is the dummy with etaDot = 0), and in the etaMass > 0 guard
// (Cu :561, :579; the Drude branch has none, Cu :605, :629).
//
// Fast path (expfac = 1 everywhere, i.e. always unless a caller has set etaDot[1] by hand).  One sub-step is
//     ed1 = ed + edd dtc/4 ;  s = exp(-dtc/2 ed1) ;  scale *= s ;  KE *= s^2 ;  eta += dtc/2 ed1 ;
//     edd = (KE - NkT)/Q ;    ed = ed1 + edd dtc/4                                   (Cu :566-585, :616-632)
// and every work-group of a rescale launch waits for S (or 2S) of them in a row, so the loop carries only what it
// must: with y = -dtc ed1 (the exponent of the KE factor) the next sub-step's y is y - (dtc^2/2) edd, so
//     KE *= exp(y) ;  edd = (KE - NkT)/Q ;  y -= (dtc^2/2) edd
// is the whole recurrence, with sum(y) carried beside it: scale = exp(sum(y)/2), eta -= sum(y)/2, and etaDot1 = -y/dtc,
// ed = ed1 - edd dtc/4 are formed once at the end.
// Same mathematics; the roundings differ from the transcription by a few ulp (parity unchanged at 1e-11).
// What of a one-link chain does not depend on the kinetic energy: formed BEFORE the wavefront waits for the sums (step_kernel:
// the index map and the launch's constants are scalar loads of kernel arguments, 1/Q is a division, expfac an exp -- half a
// microsecond of the chain wavefront's time that nobody then waits for).
struct Chain1Pre {
    Chain1Map m;
    double dtc2, dtc4, invQ0, expfac, hold, ky, c0;
    bool live, all_unit;
};
__device__ __forceinline__ Chain1Pre chain1_prepare(const ChainArgs& a, const Chain1Regs& r, const int itg) {
    Chain1Pre p;
    p.m = chain1_map(a.L, itg);
    const double dtc = a.dtc;                                        // dt / S (Cu :440-443), formed on the host
    p.dtc2 = dtc / 2.0; p.dtc4 = dtc / 4.0;
    p.live = !p.m.guard || r.etaMass > 0;
    p.invQ0 = p.live ? 1.0 / r.etaMass : 0.0;
    p.expfac = p.m.used ? chain_exp<false>(-(dtc / 8.0) * r.etaDot1) : 1.0;   // Cu :615 ; real: exp(-0) = 1
    p.all_unit = !__any(p.expfac != 1.0);
    // etaMass > 0 guard without selects: edd = KE * invQ0 + c0, with (invQ0, c0) = (1/Q, -NkT/Q) for a live thermostat
    // and (0, etaDotDot) for an inert one
    p.hold = p.live ? 0.0 : r.etaDotDot;
    p.ky = -0.5 * dtc * dtc;                                         // y' = y - dtc * 2 * dtc/4 * edd
    p.c0 = fma(-r.nkbt, p.invQ0, p.hold);
    return p;
}

__device__ __forceinline__ void chain1_finish(const ChainArgs& a, const Chain1Regs& r, const Chain1Pre& pre, double* st_out,
                                              const bool write, double* s_scale, const int itg) {
    const ChainLayout& L = a.L;
    const Chain1Map& m = pre.m;
    const double dtc = a.dtc, dtc2 = pre.dtc2, dtc4 = pre.dtc4, invQ0 = pre.invQ0, expfac = pre.expfac;
    const bool live = pre.live, all_unit = pre.all_unit;
    double ke = r.ke;
    if (write) st_out[L.off_ke + itg] = ke;                          // KE before the chain (Cu :490)
    if (!m.used) {                                                   // dualNH's middle slot: no thermostat, factor 1
        if (s_scale) s_scale[itg] = 1.0;
        if (write) {
            st_out[L.off_scale_a + itg] = 1.0; st_out[L.off_scale_b + itg] = 1.0; st_out[L.off_scale + itg] = 1.0;
            st_out[L.off_ke_post + itg] = 0.0;
        }
        return;
    }
    double ed = r.etaDot0, edd = r.etaDotDot, et = r.eta;
    const int reps = a.chain_twice ? 2 : 1;
    double total = 1.0;
    for (int rep = 0; rep < reps; rep++) {
        double scale = 1.0;
        if (live) edd = (ke - r.nkbt) * invQ0;                       // Cu :561-563, :605
        if (all_unit) {
            // One wavefront issues one fp64 instruction per ~8 cycles whether or not it depends on the one before (a
            // dependent one after 9.5: tools/micro/issue_probe.hip), so the loop is as FEW instructions as it gets -- 11,
            // where the direct transcription of Cu :566-585 has 30 and an Estrin polynomial with etaDot carried along 16:
            //   * exp(y) as the degree-6 Taylor polynomial in Horner form, 6 fmas (truncation y^7/7! < 2^-54 for
            //     |y| < 2^-6; the largest |y| seen is checked once after the loop);
            //   * edd = KE/Q + c0 with c0 = hold - NkT/Q: one fma (the etaMass > 0 guard is in the constants: (1/Q, -NkT/Q)
            //     for a live thermostat, (0, etaDotDot) for an inert one);
            //   * etaDot is not carried: y = -dtc etaDot1 throughout, so etaDot1 follows from y after the loop.
            const double ky = pre.ky, c0 = pre.c0;
            const double ed1_0 = fma(edd, dtc4, ed), ke_0 = ke;
            const double y_0 = -dtc * ed1_0;
            double y = y_0, sy = 0.0;
            TRACE(rep ? 16 : 14);
            unsigned ymax = 0u;                                      // largest |y| seen, as the high word of the double (monotone in |y|)
            // Which polynomial: the short one while every |y| stays below 2^-6 -- a thermostat near equilibrium --, chain_exp_wide
            // (|y| < 1) otherwise.  Chosen for the whole wavefront from the exponents at entry (with a factor two of room) and
            // checked after the loop: leaving the chosen range costs a second pass, it is never wrong.
            // (round 4: a middle form, degree 9 for |y| < 2^-4 in 9 fmas.  A box that starts away from equilibrium -- every fresh
            // context of bench.py, for its first ~40 steps -- sits between 2^-7 and 2^-5 and paid the wide form's 17 instructions
            // per sub-step: 197 against 190 us per launch at the metric size exactly in the window the driver's 20-step run times,
            // tools/micro/first_steps.py.)
            const unsigned ay0 = (unsigned)__double2hiint(y_0) & 0x7fffffffu;
            int form = __any(ay0 >= 0x3fa00000u) ? 2 : __any(ay0 >= 0x3f800000u) ? 1 : 0;              // some |y_0| >= 2^-5 / >= 2^-7
            bool wide = form == 2;
            if (form == 0) {
                const double k720 = 1.0 / 720.0, k120 = 1.0 / 120.0, k24 = 1.0 / 24.0, k6 = 1.0 / 6.0;
                for (int iter = 0; iter < a.S; iter++) {
                    ymax = max(ymax, (unsigned)__double2hiint(y) & 0x7fffffffu);     // two 32-bit operations (fmax on doubles: three fp64 ones)
                    double p = fma3(k720, y, k120);                      // (fma3: exactly one v_fma_f64 each -- left to itself the
                    p = fma3(p, y, k24);                                 //  compiler copies every constant into the accumulator of a
                    p = fma3(p, y, k6);                                  //  two-address fmac first: four more fp64-rate moves per sub-step)
                    p = fma(p, y, 0.5);
                    p = fma(p, y, 1.0);
                    p = fma(p, y, 1.0);
                    ke *= p;                                             // Cu :574, :621
                    sy += y;
                    edd = fma3(ke, invQ0, c0);                           // Cu :579-581, :629
                    y = fma3(edd, ky, y);                                // Cu :583-585 / :630-632 and the next :568-570
                }
                if (__builtin_expect(__any(ymax >= 0x3f900000u), 0)) {   // some |y| >= 2^-6: again, with the next polynomial
                    ke = ke_0; y = y_0; sy = 0.0; ymax = 0u; form = 1;
                }
            }
            if (form == 1) {
                for (int iter = 0; iter < a.S; iter++) {
                    ymax = max(ymax, (unsigned)__double2hiint(y) & 0x7fffffffu);
                    ke *= chain_exp9(y);
                    sy += y;
                    edd = fma3(ke, invQ0, c0);
                    y = fma3(edd, ky, y);
                }
                if (__builtin_expect(__any(ymax >= 0x3fb00000u), 0)) {   // some |y| >= 2^-4: again, with the long polynomial
                    ke = ke_0; y = y_0; sy = 0.0; ymax = 0u; wide = true;
                }
            }
            if (wide) {
                for (int iter = 0; iter < a.S; iter++) {
                    ymax = max(ymax, (unsigned)__double2hiint(y) & 0x7fffffffu);
                    ke *= chain_exp_wide(y);
                    sy += y;
                    edd = fma(ke, invQ0, c0);
                    y = fma(edd, ky, y);
                }
                if (__builtin_expect(__any(ymax >= 0x3ff00000u), 0)) {   // some |y| >= 1: again, with the range reduction
                    ke = ke_0; y = y_0; sy = 0.0;
                    for (int iter = 0; iter < a.S; iter++) {
                        ke *= chain_exp<false>(y);
                        sy += y;
                        edd = fma(ke, invQ0, c0);
                        y = fma(edd, ky, y);
                    }
                }
            }
            TRACE(rep ? 11 : 16);
            const double ed1 = fma(y - y_0, -a.inv_dtc, ed1_0);      // etaDot1 advanced by the same kicks as y
            ed = fma(edd, -dtc4, ed1);                               // ed1 is one quarter-kick ahead
            scale = chain_exp<false>(0.5 * sy);                      // prod exp(-dtc/2 ed1)      (Cu :573, :620)
            et = fma(-0.5, sy, et);                                  // sum dtc/2 ed1            (Cu :575-577, :623)
        } else {
            const double ef2 = expfac * expfac, efd = expfac * dtc4;
            for (int iter = 0; iter < a.S; iter++) {
                ed = expfac == 1.0 ? fma(edd, dtc4, ed) : fma(ed, ef2, edd * efd);      // (ed*ef + edd*dtc4)*ef
                const double e = chain_exp<false>(-dtc2 * ed);
                scale *= e; ke *= e * e;
                et = fma(dtc2, ed, et);
                if (live) edd = (ke - r.nkbt) * invQ0;
                ed = expfac == 1.0 ? fma(edd, dtc4, ed) : fma(ed, ef2, edd * efd);
            }
        }
        if (write) {
            if (rep == 0) { st_out[L.off_scale_a + itg] = scale; st_out[L.off_ke_post + itg] = ke; }
            else st_out[L.off_scale_b + itg] = scale;
        }
        total *= scale;
    }
    if (s_scale) s_scale[itg] = total;
    TRACE(12);
    if (!write) return;
    if (reps == 1) st_out[L.off_scale_b + itg] = 1.0;
    st_out[L.off_scale + itg] = total;
    st_out[L.off_eta + m.eta] = et;
    st_out[L.off_etaDotDot + m.edd] = edd;
    st_out[L.off_etaDot + m.ed0] = ed;
    st_out[L.off_etaDot + m.ed1] = r.etaDot1;
}

__device__ __forceinline__ void chain1_run(const ChainArgs& a, const Chain1Regs& r, double* st_out, const bool write,
                                           double* s_scale, const int itg) {
    chain1_finish(a, r, chain1_prepare(a, r, itg), st_out, write, s_scale, itg);
}

// dualNH, one link, useDrudeNHChains = false (the C++ default).  Ref :139-154 then leaves numTempGroup = 1, so the
// descending loop (Ref :476-481) damps link i with etaDot[i+1]: the Drude thermostat (vector index 1) with the first
// dummy, the REAL thermostat (index 0) with the Drude thermostat's etaDot -- the value just updated in the same
// sub-step -- while the ascending loop (Ref :494-503, stride 2 hard-coded) damps both with their dummies.  Same
// storage as the self-consistent layout (Chain1Map); lanes 0 (real) and 2 (Drude) of one wavefront, one shuffle per
// sub-step carries the coupling; lane 1 is the unused middle slot.  The arithmetic is run_dualnh<1>'s.
__device__ __forceinline__ void chain1q_run(const ChainArgs& a, const Chain1Regs& r, double* st_out, const bool write,
                                            double* s_scale, const int itg) {
    const ChainLayout& L = a.L;
    const Chain1Map m = chain1_map(L, itg);
    const bool drude = itg == 2;
    const double dtc = a.dt / a.S;                                   // Ref :432-435
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    double ke = r.ke;
    if (write) st_out[L.off_ke + itg] = ke;
    if (!m.used) {
        if (s_scale) s_scale[itg] = 1.0;
        if (write) {
            st_out[L.off_scale_a + itg] = 1.0; st_out[L.off_scale_b + itg] = 1.0; st_out[L.off_scale + itg] = 1.0;
            st_out[L.off_ke_post + itg] = 0.0;
        }
        return;
    }
    const double invQ = 1.0 / r.etaMass;                             // Ref :471-472 divides unconditionally
    const double dummy2 = __shfl(r.etaDot1, 0, 64);                  // etaDot[2]: the real lane's "link 1"
    const double efD1 = chain_exp<false>(-dtc8 * dummy2);            // Drude, descending loop: exp(-dtc8 etaDot[1+1])
    const double ef2 = chain_exp<false>(-dtc8 * r.etaDot1);          // ascending loop: exp(-dtc8 etaDot[i+2]), i = 0 / 1
    double ed = r.etaDot0, edd = r.etaDotDot, et = r.eta;
    const int reps = a.chain_twice ? 2 : 1;
    double total = 1.0;
    for (int rep = 0; rep < reps; rep++) {
        double scale = 1.0;
        edd = (ke - r.nkbt) * invQ;                                  // Ref :471-472
        for (int iter = 0; iter < a.S; iter++) {
            double t1 = ed * efD1;                                   // Ref :476-481, i = 1 (Drude) first ...
            t1 = fma(edd, dtc4, t1);
            t1 *= efD1;
            const double efr = chain_exp<false>(-dtc8 * __shfl(t1, 2, 64));   // ... then i = 0 with the Drude etaDot just formed
            double t0 = ed * efr;
            t0 = fma(edd, dtc4, t0);
            t0 *= efr;
            ed = drude ? t1 : t0;
            const double e = chain_exp<false>(-dtc2 * ed);           // Ref :483-486
            scale *= e; ke *= e * e;
            et = fma(dtc2, ed, et);                                  // Ref :487-489
            edd = (ke - r.nkbt) * invQ;                              // Ref :491-492
            ed *= ef2;                                               // Ref :494-503
            ed = fma(edd, dtc4, ed);
            ed *= ef2;
        }
        if (write) {
            if (rep == 0) { st_out[L.off_scale_a + itg] = scale; st_out[L.off_ke_post + itg] = ke; }
            else st_out[L.off_scale_b + itg] = scale;
        }
        total *= scale;
    }
    if (s_scale) s_scale[itg] = total;
    if (!write) return;
    if (reps == 1) st_out[L.off_scale_b + itg] = 1.0;
    st_out[L.off_scale + itg] = total;
    st_out[L.off_eta + m.eta] = et;
    st_out[L.off_etaDotDot + m.edd] = edd;
    st_out[L.off_etaDot + m.ed0] = ed;
    st_out[L.off_etaDot + m.ed1] = r.etaDot1;
}

// dualNH WITHOUT useDrudeNHChains (the C++ default; Ref :139-154 leaves numTempGroup = 1) and 2-4 links: the vectors are
// [real0, drude0, real1, ..., real(C-1) | dummy, dummy] and the loops of Ref :476-481 / :494-503 make ONE coupled chain of C + 1
// moving entries out of them -- the descending loop damps entry i with entry i + 1 (real0 with the Drude thermostat's etaDot, the
// Drude thermostat with real1's), the ascending loop with entry i + 2, and the odd entries from 3 on take drudekbT (Ref :498).  The
// same sequence of operations as the transcription below, in the form of the other fast chains: entries in registers, exp() a
// short polynomial without a range test inside the loop (chain_exp9 / chain_exp_wide, the largest argument checked afterwards),
// the two dummies' factors -- constants of the call -- formed once, (...)/etaMass[i] as a multiplication by 1/Q: 2 C + 1
// polynomials per sub-step where the transcription evaluates 2 C + 4 range-tested exponentials.  False: an argument left every
// fast form's range (nothing has been written; the caller runs the transcription).
template <int CC, bool WIDE>
__device__ __forceinline__ double dualnh_quirk_loop(double* eta, double* ed, double* edd, const double* em, const double* invM,
                                                    const double dtc2, const double dtc4, const double dtc8, const int S,
                                                    const double realNkbT, const double drudeNkbT, const double realkbT, const double drudekbT,
                                                    const double ef_d1, const double ef_d2, double& realKE, double& drudeKE, double& sR, double& sD) {
    constexpr int N = CC + 1;                                        // moving entries 0 .. C; ed[N], ed[N + 1] are the dummies
    double xmax = 0.0;
    auto ex = [&](const double x) { xmax = fmax(xmax, fabs(x)); return WIDE ? chain_exp_wide(x) : chain_exp9(x); };
    for (int iter = 0; iter < S; iter++) {
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {                           // Ref :476-481, i = idxMaxNHChains .. 0, stride numTempGroup = 1
            const double ef = i == N - 1 ? ef_d1 : ex(-dtc8 * ed[i + 1]);
            ed[i] *= ef; ed[i] += edd[i] * dtc4; ed[i] *= ef;
        }
        { const double e = ex(-dtc2 * ed[0]); sR *= e; realKE *= e * e; }        // Ref :483-486
        { const double e = ex(-dtc2 * ed[1]); sD *= e; drudeKE *= e * e; }
#pragma unroll
        for (int i = 0; i < N; i++) eta[i] += dtc2 * ed[i];          // Ref :487-489
        edd[0] = (realKE - realNkbT) * invM[0];                      // Ref :491-492
        edd[1] = (drudeKE - drudeNkbT) * invM[1];
#pragma unroll
        for (int i = 0; i < N; i++) {                                // Ref :494-503, stride 2 hard-coded
            const double ef = i == N - 1 ? ef_d2 : (i == N - 2 ? ef_d1 : ex(-dtc8 * ed[i + 2]));
            ed[i] *= ef;
            if (i > 1) edd[i] = (em[i - 2] * ed[i - 2] * ed[i - 2] - (i % 2 == 0 ? realkbT : drudekbT)) * invM[i];
            ed[i] += edd[i] * dtc4; ed[i] *= ef;
        }
    }
    return xmax;
}
template <int CC, bool LIBM>
__device__ __forceinline__ bool dualnh_quirk_fast(const ChainArgs& a, const double* st_in, double* st_out, const bool write,
                                                  double* s_scale, const double ke0, const double ke1, const double ke2) {
    const ChainLayout& L = a.L;
    constexpr int N = CC + 1;
    double eta[N], ed[N + 2], edd[N], em[N], invM[N];
#pragma unroll
    for (int i = 0; i < N; i++) { eta[i] = st_in[L.off_eta + i]; edd[i] = st_in[L.off_etaDotDot + i]; em[i] = st_in[L.off_etaMass + i]; invM[i] = 1.0 / em[i]; }
#pragma unroll
    for (int i = 0; i < N + 2; i++) ed[i] = st_in[L.off_etaDot + i];
    const double realNkbT = st_in[L.off_nkbt + 0], drudeNkbT = st_in[L.off_nkbt + 2];
    const double dtc = a.dt / a.S;                                   // Ref :432-435
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    const double ef_d1 = chain_exp<LIBM>(-dtc8 * ed[N]), ef_d2 = chain_exp<LIBM>(-dtc8 * ed[N + 1]);    // the dummies never move
    double realKE = ke0, drudeKE = ke2;
    const int reps = a.chain_twice ? 2 : 1;
    double sc0R = 1.0, sc0D = 1.0, sc1R = 1.0, sc1D = 1.0, ke_post0 = 0.0, ke_post2 = 0.0;     // (no array indexed by rep: that would live on the stack)
    for (int rep = 0; rep < reps; rep++) {
        edd[0] = (realKE - realNkbT) * invM[0];                      // Ref :471-472
        edd[1] = (drudeKE - drudeNkbT) * invM[1];
        double x0 = fmax(fabs(dtc2 * ed[0]), fabs(dtc2 * ed[1]));
#pragma unroll
        for (int i = 1; i < N; i++) x0 = fmax(x0, fabs(dtc8 * ed[i]));
        // the form is chosen from the exponents at entry with a factor two of room and checked afterwards.  Short chains keep this
        // rep's entry state in registers and start again with the wide form when the narrow one was left; long ones (5-16 links: the
        // copies would be another 3 x 17 doubles) hand over to the transcription instead
        constexpr bool SAVE = CC <= 4;
        constexpr int NS = SAVE ? N : 1;
        double s_eta[NS], s_ed[NS], s_edd[NS];
        if constexpr (SAVE) {
#pragma unroll
            for (int i = 0; i < N; i++) { s_eta[i] = eta[i]; s_ed[i] = ed[i]; s_edd[i] = edd[i]; }
        }
        const double kr = realKE, kd = drudeKE;
        double sR = 1.0, sD = 1.0, xmax = 0.0;
        bool wide = x0 >= 0.03125;
        if (!wide) {
            xmax = dualnh_quirk_loop<CC, false>(eta, ed, edd, em, invM, dtc2, dtc4, dtc8, a.S, realNkbT, drudeNkbT, a.realkbT, a.drudekbT, ef_d1, ef_d2, realKE, drudeKE, sR, sD);
            if (xmax >= 0.0625) {
                if constexpr (!SAVE) return false;
#pragma unroll
                for (int i = 0; i < NS; i++) { eta[i] = s_eta[i]; ed[i] = s_ed[i]; edd[i] = s_edd[i]; }
                realKE = kr; drudeKE = kd; sR = 1.0; sD = 1.0;
                wide = true;
            }
        }
        if (wide) {
            xmax = dualnh_quirk_loop<CC, true>(eta, ed, edd, em, invM, dtc2, dtc4, dtc8, a.S, realNkbT, drudeNkbT, a.realkbT, a.drudekbT, ef_d1, ef_d2, realKE, drudeKE, sR, sD);
            if (xmax >= 1.0) return false;
        }
        if (rep == 0) { sc0R = sR; sc0D = sD; ke_post0 = realKE; ke_post2 = drudeKE; } else { sc1R = sR; sc1D = sD; }
    }
    const double totR = sc0R * sc1R, totD = sc0D * sc1D;
    if (s_scale) { s_scale[0] = totR; s_scale[1] = 1.0; s_scale[2] = totD; }
    if (!write) return true;
    st_out[L.off_ke + 0] = ke0; st_out[L.off_ke + 1] = ke1; st_out[L.off_ke + 2] = ke2;
    st_out[L.off_kesum] = 0.5 * (ke0 + ke2);
    st_out[L.off_scale_a + 0] = sc0R; st_out[L.off_scale_a + 1] = 1.0; st_out[L.off_scale_a + 2] = sc0D;
    st_out[L.off_ke_post + 0] = ke_post0; st_out[L.off_ke_post + 1] = 0.0; st_out[L.off_ke_post + 2] = ke_post2;
    st_out[L.off_scale_b + 0] = sc1R; st_out[L.off_scale_b + 1] = 1.0; st_out[L.off_scale_b + 2] = sc1D;
    st_out[L.off_scale + 0] = totR; st_out[L.off_scale + 1] = 1.0; st_out[L.off_scale + 2] = totD;
#pragma unroll
    for (int i = 0; i < N; i++) { st_out[L.off_eta + i] = eta[i]; st_out[L.off_etaDotDot + i] = edd[i]; }
#pragma unroll
    for (int i = 0; i < N + 2; i++) st_out[L.off_etaDot + i] = ed[i];
    return true;
}

// The Reference platform's coupled real/Drude chain on its interleaved vectors.  Ref :467-504.
// LEN = compile-time bound of the vectors (2*CC+2), 0 = dynamic (LDS).
template <int CC, bool LIBM = true, bool QUIRK_FAST = true>
__device__ __forceinline__ void run_dualnh(const ChainArgs& a, const double* st_in, double* st_out, const bool write,
                                           double* s_scale, double* lds, const double ke0, const double ke1, const double ke2) {
    const ChainLayout& L = a.L;
    if constexpr (CC >= 2 && QUIRK_FAST) {                           // the coupled chain of the C++ default through the fast forms first
        if (L.numTempGroup == 1 && dualnh_quirk_fast<CC, LIBM>(a, st_in, st_out, write, s_scale, ke0, ke1, ke2)) return;
    }
    constexpr int LM = CC > 0 ? 2 * CC + 2 : 1;
    double r_eta[LM], r_etaDot[LM], r_etaDotDot[LM], r_etaMass[LM];
    double *eta = r_eta, *etaDot = r_etaDot, *etaDotDot = r_etaDotDot, *etaMass = r_etaMass;
    const int n = L.len_eta, nd = L.len_etaDot;                      // n = 2C or C+1 ; nd = n+2
    if (CC == 0) { eta = lds; etaDot = eta + nd; etaDotDot = etaDot + nd; etaMass = etaDotDot + nd; }
    const int NB = CC > 0 ? LM : nd;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        eta[i] = i < n ? st_in[L.off_eta + i] : 0.0;
        etaDotDot[i] = i < n ? st_in[L.off_etaDotDot + i] : 0.0;
        etaMass[i] = i < n ? st_in[L.off_etaMass + i] : 1.0;
        etaDot[i] = i < nd ? st_in[L.off_etaDot + i] : 0.0;
    }
    const double realNkbT = st_in[L.off_nkbt + 0], drudeNkbT = st_in[L.off_nkbt + 2];
    const double dtc = a.dt / a.S;                                   // Ref :432-435
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    const int ntg = L.numTempGroup, idxMax = L.idxMaxNHChains, iNum = L.iNumNHChains;
    double realKE = ke0, drudeKE = ke2;
    if (write) {
        st_out[L.off_ke + 0] = realKE; st_out[L.off_ke + 1] = ke1; st_out[L.off_ke + 2] = drudeKE;
        st_out[L.off_kesum] = 0.5 * (realKE + drudeKE);
    }
    const int reps = a.chain_twice ? 2 : 1;
    double totR = 1.0, totD = 1.0;
    for (int rep = 0; rep < reps; rep++) {
        double scaleReal = 1.0, scaleDrude = 1.0, expfac = 1.0;
        const double invQr = 1.0 / etaMass[0], invQd = 1.0 / etaMass[1];
        etaDotDot[0] = (realKE - realNkbT) * invQr;                  // Ref :471-472
        etaDotDot[1] = (drudeKE - drudeNkbT) * invQd;
        for (int iter = 0; iter < a.S; iter++) {
#pragma unroll
            for (int i = NB - 3; i >= 0; i--) {                      // Ref :476-481 (i = idxMaxNHChains .. 0)
                if (i <= idxMax) {
                    expfac = chain_exp<LIBM>(-dtc8 * (ntg == 2 ? etaDot[i + 2] : etaDot[i + 1]));
                    etaDot[i] *= expfac;
                    etaDot[i] += etaDotDot[i] * dtc4;
                    etaDot[i] *= expfac;
                }
            }
            { const double e = chain_exp<LIBM>(-dtc2 * etaDot[0]); scaleReal *= e; realKE *= e * e; }    // Ref :483-486
            { const double e = chain_exp<LIBM>(-dtc2 * etaDot[1]); scaleDrude *= e; drudeKE *= e * e; }
#pragma unroll
            for (int i = 0; i < NB - 2; i++) if (i < iNum) eta[i] += dtc2 * etaDot[i];             // Ref :487-489
            etaDotDot[0] = (realKE - realNkbT) * invQr;              // Ref :491-492
            etaDotDot[1] = (drudeKE - drudeNkbT) * invQd;
#pragma unroll
            for (int i = 0; i < NB - 2; i++) {                       // Ref :494-503
                if (i < iNum) {
                    expfac = chain_exp<LIBM>(-dtc8 * etaDot[i + 2]);
                    etaDot[i] *= expfac;
                    if (i > 1) {
                        const double dofkbT = (i % 2 == 0 ? a.realkbT : a.drudekbT);
                        etaDotDot[i] = (etaMass[i - 2] * etaDot[i - 2] * etaDot[i - 2] - dofkbT) / etaMass[i];
                    }
                    etaDot[i] += etaDotDot[i] * dtc4;
                    etaDot[i] *= expfac;
                }
            }
        }
        if (write) {
            if (rep == 0) {
                st_out[L.off_scale_a + 0] = scaleReal; st_out[L.off_scale_a + 1] = 1.0; st_out[L.off_scale_a + 2] = scaleDrude;
                st_out[L.off_ke_post + 0] = realKE; st_out[L.off_ke_post + 1] = 0.0; st_out[L.off_ke_post + 2] = drudeKE;
            } else {
                st_out[L.off_scale_b + 0] = scaleReal; st_out[L.off_scale_b + 1] = 1.0; st_out[L.off_scale_b + 2] = scaleDrude;
            }
        }
        totR *= scaleReal; totD *= scaleDrude;
    }
    if (s_scale) { s_scale[0] = totR; s_scale[1] = 1.0; s_scale[2] = totD; }
    if (!write) return;
    if (reps == 1) { st_out[L.off_scale_b + 0] = 1.0; st_out[L.off_scale_b + 1] = 1.0; st_out[L.off_scale_b + 2] = 1.0; }
    st_out[L.off_scale + 0] = totR; st_out[L.off_scale + 1] = 1.0; st_out[L.off_scale + 2] = totD;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        if (i < n) { st_out[L.off_eta + i] = eta[i]; st_out[L.off_etaDotDot + i] = etaDotDot[i]; }
        if (i < nd) st_out[L.off_etaDot + i] = etaDot[i];
    }
}

// ---- chains of 5-16 links (TGNH): the links of a thermostat across the lanes of a 16-lane row --------------------------------
// run_tgnh<0> keeps such a chain in LDS (dynamic indexing rules registers out): every access is a ~100-cycle round trip on a
// path that is serial by nature -- ~180 us per time step for the ten links of the reference's own test
// (TestReferenceDrudeTGNHIntegrator.cpp:166).  Here lane l of a row holds link l of the row's thermostat in REGISTERS and the
// sweeps of Cu :566-571 / :586-592 run as C steps in which every lane forms its candidate update from its neighbour's current
// value (one DPP row shift) and the lane whose turn it is commits it: same arithmetic, same order per link, no memory on the
// path.  16 thermostats per 256-thread work-group and batch.  exp(-dtc/8 etaDot[i+1]) of the ascending sweep is the factor
// the descending sweep left in the lane (etaDot[i+1] has not moved in between): taken once, as in chain_real_core<CC >= 2>.
template <int CTRL> __device__ __forceinline__ double chain_dpp(const double x) {      // row_shl:1 = 0x101 (lane + 1), row_shr:1 = 0x111 (lane - 1); 0 at a row's end
    const int xl = __double2loint(x), xh = __double2hiint(x);
    const int lo = __builtin_amdgcn_update_dpp(xl, xl, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(xh, xh, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void chain_lanes_run(const ChainArgs& a, double* st, const int tid, const int nthreads, const double* s_ke) {
    const ChainLayout& L = a.L;
    const int C = L.C, NT = L.NT, link = tid & 15;
    const double dtc = a.dt / a.S;                                   // Cu :440-443
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    const int reps = a.chain_twice ? 2 : 1;
    for (int t0 = 0; t0 < NT; t0 += nthreads / 16) {                 // (uniform trip count: every lane runs every batch)
        const int t = t0 + (tid >> 4);
        const bool valid = t < NT && link < C;
        const bool drude = t == NT - 1;
        const int tt = t < NT ? t : NT - 1;                          // (idle rows shadow the last thermostat, store nothing)
        const int li = link < C ? link : C - 1;
        double eta = st[L.off_eta + tt * C + li], ed = st[L.off_etaDot + tt * (C + 1) + li], edd = st[L.off_etaDotDot + tt * C + li];
        const double Q = st[L.off_etaMass + tt * C + li], dummy = st[L.off_etaDot + tt * (C + 1) + C];
        const double nkbt = st[L.off_nkbt + tt], kbT = drude ? a.drudekbT : a.realkbT;
        const bool live = drude || Q > 0;                            // (asked of link 0: Cu :561 vs :605)
        const int nmove = (drude && !L.use_drude_chains) ? 1 : C;    // Cu :607-614, :633-641: the Drude thermostat's higher links only with useDrudeNHChains
        const bool mine = valid && link < nmove;
        const double invQ = Q > 0 ? 1.0 / Q : 0.0;
        const double Qprev = chain_dpp<0x111>(Q);
        double ke = s_ke[tt];
        if (valid && link == 0) st[L.off_ke + t] = ke;               // KE before the chain (Cu :490)
        double total = 1.0, ef = 1.0;
        for (int rep = 0; rep < reps; rep++) {
            double scale = 1.0;
            if (link == 0 && live) edd = (ke - nkbt) * invQ;         // Cu :561-563, :605
            for (int iter = 0; iter < a.S; iter++) {
                for (int i = C - 1; i >= 0; i--) {                   // Cu :566-571 / :607-618
                    const double nb = chain_dpp<0x101>(ed);
                    const double efn = chain_exp<false>(-dtc8 * (link == C - 1 ? dummy : nb));
                    double cand = ed * efn;
                    cand += edd * dtc4;
                    cand *= efn;
                    if (mine && link == i) { ed = cand; ef = efn; }
                }
                {                                                    // Cu :573-577 / :620-623
                    const double e = chain_exp<false>(-dtc2 * ed);
                    if (link == 0) { scale *= e; ke *= e * e; }
                }
                if (mine) eta += dtc2 * ed;
                if (link == 0 && live) edd = (ke - nkbt) * invQ;     // Cu :579-581 / :629
                if (mine && link == 0) { ed *= ef; ed += edd * dtc4; ed *= ef; }       // Cu :583-585 / :630-632 (link 0's expfac reused)
                for (int i = 1; i < C; i++) {                        // Cu :586-592 / :633-641
                    const double prev = chain_dpp<0x111>(ed);
                    if (mine && link == i) {
                        ed *= ef;
                        edd = (Qprev * prev * prev - kbT) * invQ;
                        ed += edd * dtc4;
                        ed *= ef;
                    }
                }
            }
            if (valid && link == 0) {
                if (rep == 0) { st[L.off_scale_a + t] = scale; st[L.off_ke_post + t] = ke; }
                else st[L.off_scale_b + t] = scale;
            }
            total *= scale;
        }
        if (valid) {
            st[L.off_eta + t * C + link] = eta; st[L.off_etaDot + t * (C + 1) + link] = ed; st[L.off_etaDotDot + t * C + link] = edd;
            if (link == 0) {
                if (reps == 1) st[L.off_scale_b + t] = 1.0;
                st[L.off_scale + t] = total;
            }
        }
    }
}

// dualNH with useDrudeNHChains, CC = 2-4 links, inside a streaming launch: on the Reference platform's interleaved vectors
// [real0, drude0, real1, drude1, ...] link i couples to link i + 2 (Ref :477, :495, numTempGroup = 2), i.e. the real chain and the Drude
// chain never meet -- two independent chains with the arithmetic of the TGNH ones (SURVEY A9's bridge identity; Ref :471 divides
// without the etaMass > 0 guard: both as TGNH's Drude thermostat).  Lanes 0 (real) and 2 (Drude) of the internal [real, unused,
// Drude] order take one each through chain_both_fast, in one pass; run_dualnh, the transcription, ran them in lane 0 one after the
// other with 2C range-tested exponentials per sub-step.  Called by lanes 0 and 2 together.  False (nothing written): the
// arguments left the fast forms' range -- the caller falls back to run_dualnh.
template <int CC>
__device__ __forceinline__ bool run_dualnh_pair(const ChainArgs& a, const double* st_in, double* st_out, const bool write,
                                                double* s_scale, const int itg, const double ke0, const double ke1, const double ke2) {
    const ChainLayout& L = a.L;
    const int tt = itg >> 1;                                         // 0 the real chain, 1 the Drude chain
    double eta[CC], etaDot[CC + 1], etaDotDot[CC], etaMass[CC];
#pragma unroll
    for (int i = 0; i < CC; i++) {
        eta[i] = st_in[L.off_eta + 2 * i + tt]; etaDotDot[i] = st_in[L.off_etaDotDot + 2 * i + tt]; etaMass[i] = st_in[L.off_etaMass + 2 * i + tt];
    }
#pragma unroll
    for (int i = 0; i <= CC; i++) etaDot[i] = st_in[L.off_etaDot + 2 * i + tt];     // (len_etaDot = 2 C + 2: the two dummies, Ref :216-217)
    ChainConst k;
    const double dtc = a.dt / a.S;                                   // Ref :432-435
    k.dtc2 = dtc / 2.0; k.dtc4 = dtc / 4.0; k.dtc8 = dtc / 8.0; k.S = a.S;
    const double nkbt = st_in[L.off_nkbt + itg], kbT = tt ? a.drudekbT : a.realkbT;
    const int reps = a.chain_twice ? 2 : 1;
    double ke = tt ? ke2 : ke0, sc0 = 1.0, sc1 = 1.0, ke_post = 0.0;
    for (int rep = 0; rep < reps; rep++) {
        double kep, sc_rep;
        if (!chain_both_fast<CC, false>(eta, etaDot, etaDotDot, etaMass, k, nkbt, kbT, true, ke, &sc_rep, &kep)) return false;
        if (rep == 0) { sc0 = sc_rep; ke_post = kep; } else sc1 = sc_rep;      // (no array indexed by rep: that would live on the stack)
        ke = kep;
    }
    const double total = sc0 * sc1;
    if (s_scale) { s_scale[itg] = total; if (itg == 0) s_scale[1] = 1.0; }
    if (!write) return true;
    if (itg == 0) {                                                  // what run_dualnh writes for all three slots at once
        st_out[L.off_ke + 0] = ke0; st_out[L.off_ke + 1] = ke1; st_out[L.off_ke + 2] = ke2;
        st_out[L.off_kesum] = 0.5 * (ke0 + ke2);
        st_out[L.off_scale_a + 1] = 1.0; st_out[L.off_scale_b + 1] = 1.0; st_out[L.off_scale + 1] = 1.0; st_out[L.off_ke_post + 1] = 0.0;
    }
    st_out[L.off_scale_a + itg] = sc0; st_out[L.off_ke_post + itg] = ke_post;
    st_out[L.off_scale_b + itg] = sc1;                             // (1 when the chain ran once)
    st_out[L.off_scale + itg] = total;
#pragma unroll
    for (int i = 0; i < CC; i++) { st_out[L.off_eta + 2 * i + tt] = eta[i]; st_out[L.off_etaDotDot + 2 * i + tt] = etaDotDot[i]; }
#pragma unroll
    for (int i = 0; i <= CC; i++) st_out[L.off_etaDot + 2 * i + tt] = etaDot[i];
    return true;
}

// Chains of 2-4 links inside a streaming launch (the one-link chains have chain1_run): called by the 64 lanes of one
// wavefront, converged; lane itg < NT holds its thermostat's summed kinetic energy `ke`.  TGNH: lane itg runs its thermostat
// (run_tgnh, register-resident links; the real thermostats and the Drude thermostat in ONE pass through the fast forms,
// chain_both_fast -- until round 4 they were two code paths of this wavefront, one after the other, and the serial section every
// work-group waits for was twice as long: C2 with three links 33.6 k -> 44.5 k steps/s).  dualNH: lanes 0 and 2 run the Reference platform's two chains where they are independent (run_dualnh_pair), lane 0 its coupled vectors otherwise (run_dualnh).  No library exp
// (chain_exp<false>): ocml's would cost the streaming kernels ~30 registers.  Longer chains keep their own launch (chain_kernel).
// QUIRK_FAST = false (wstep_kernel, at its 228 registers): the coupled dualNH chain (useDrudeNHChains = false) stays the transcription
// there -- with the fast form beside it the kernel spills.
template <bool QUIRK_FAST = true>
__device__ __forceinline__ void chainN_run(const ChainArgs& a, const double* st_in, double* st_out, const bool write,
                                           double* s_scale, const int itg, const double ke) {
    const ChainLayout& L = a.L;
    if (L.mode == TGNH_MODE_TGNH) {
        if (itg < L.NT) {
            switch (L.C) {
                case 2: run_tgnh<2, false, true>(a, st_in, st_out, write, s_scale, itg, nullptr, ke); break;
                case 3: run_tgnh<3, false, true>(a, st_in, st_out, write, s_scale, itg, nullptr, ke); break;
                default: run_tgnh<4, false, true>(a, st_in, st_out, write, s_scale, itg, nullptr, ke); break;
            }
        }
    } else {
        const double ke0 = __shfl(ke, 0, 64), ke1 = __shfl(ke, 1, 64), ke2 = __shfl(ke, 2, 64);
        bool done = false;
        if (L.use_drude_chains != 0) {                               // two independent chains: lanes 0 and 2, one pass (run_dualnh_pair)
            bool ok = false;
            if (itg == 0 || itg == 2) {
                switch (L.C) {
                    case 2: ok = run_dualnh_pair<2>(a, st_in, st_out, write, s_scale, itg, ke0, ke1, ke2); break;
                    case 3: ok = run_dualnh_pair<3>(a, st_in, st_out, write, s_scale, itg, ke0, ke1, ke2); break;
                    default: ok = run_dualnh_pair<4>(a, st_in, st_out, write, s_scale, itg, ke0, ke1, ke2); break;
                }
            }
            done = __shfl((int)ok, 0, 64) != 0;                      // (the same answer in both lanes: chain_fast votes)
        }
        if (!done && itg == 0) {
            switch (L.C) {
                case 2: run_dualnh<2, false, QUIRK_FAST>(a, st_in, st_out, write, s_scale, nullptr, ke0, ke1, ke2); break;
                case 3: run_dualnh<3, false, QUIRK_FAST>(a, st_in, st_out, write, s_scale, nullptr, ke0, ke1, ke2); break;
                default: run_dualnh<4, false, QUIRK_FAST>(a, st_in, st_out, write, s_scale, nullptr, ke0, ke1, ke2); break;
            }
        }
    }
}

#pragma clang fp contract(fast)

}  // namespace tgnh
#endif
