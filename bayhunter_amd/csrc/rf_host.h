// rf_host.h -- host-side preparation of the RF launch constants and the FFT twiddle table.
// Everything here is evaluated with the reference's own expressions (cited) on the host, so the
// device consumes bit-identical constants.
#pragma once
#include <cmath>
#include <complex>
#include "rf_core.h"

namespace bh {

constexpr double RF_WEIGHT_CUTOFF = 3.0e-19;

inline void rf_fill_launch(RfLaunch &P, double p, double gauss, int nsamp, double fsamp,
                           double tshift, double nsv, int waveno, int nout)
{
    P.slowness = p * 0.00899;                                  // wrap.cpp:55,76
    P.p2 = P.slowness * P.slowness;                            // greens.cpp:421
    P.gauss = gauss;
    P.tshift = tshift;
    P.nsv = nsv;
    P.dw = 2.0 * M_PI * fsamp / nsamp;                         // greens.cpp:360,507
    P.qgauss = std::sqrt(M_PI) * fsamp / gauss;                // greens.cpp:361
    P.sc = std::sqrt(1. / (double)nsamp);                      // fork.cpp:28
    P.qn = 1. / std::sqrt((double)nsamp);                      // greens.cpp:147
    P.wref = 2. * M_PI * 1.;                                   // greens.cpp:447 with fref = 1 (synrf.cpp:25)
    P.nsamp = nsamp;
    P.nfreq = nsamp / 2 + 1;
    P.log2n = 0;
    while ((1 << P.log2n) < nsamp) P.log2n++;
    P.waveno = waveno;
    P.nout = nout;
    // The spectrum is multiplied by exp(-(w/a)^2/4) (greens.cpp:389-392).  Frequencies whose weight is
    // below RF_WEIGHT_CUTOFF of its value at w = 0 are set to zero instead of being computed: their
    // contribution to a sample is cutoff x |R/Z| (the deconvolved spectral ratio, O(1..1e3) for a
    // layered model with attenuation), i.e. <= 1e-16 of the trace's scale -- five orders below the
    // parity tolerance of 1e-10 even for a ratio of 1e4 (summed over the ~44 dropped bins: 4e-15).  Checked
    // on the models most likely to break it -- thin very slow surface layer, Q down to 5, a = 0.8..1.2 --
    // against the oracle, which computes every bin: <= 2.9e-15 (tests/rf_extreme.py, test_hostsim.py and the
    // GPU tier).  a = 1, 5 Hz, nsamp 512: 213 of 257
    // frequencies are computed (5 instead of 6 passes of a 256-thread workgroup over 6 models; the
    // round-1 cutoff of 1e-24 kept 243); a >= 1.21: all of them.
    const double wcut = 2.0 * gauss * std::sqrt(std::log(1.0 / RF_WEIGHT_CUTOFF));
    double jcut = std::floor(wcut / P.dw) + 1.0;
    P.nact = (jcut < (double)P.nfreq) ? (int)jcut : P.nfreq;
    if (P.nact < 1) P.nact = 1;
}

// Per-frequency constants of phase 3 (rf_core.h, RfFreq), with the reference's expressions:
// tab[3j] = ln(w/wref) (0 for j = 0, greens.cpp:530), tab[3j+1], tab[3j+2] = q exp(Complex(-(w/a)^2/4,
// -w tshift)) with w/a capped at 50 (greens.cpp:389-392), w = dw j.
inline void rf_fill_freq_table(const RfLaunch &P, double *tab)
{
    for (int j = 0; j < P.nfreq; j++) {
        const double w = P.dw * j;
        tab[RF_FTAB * j] = j ? std::log(w / P.wref) : 0.0;
        double wa = w / P.gauss;
        wa = (wa > 50.0) ? 50.0 : wa;
        const std::complex<double> cq = P.qgauss * std::exp(std::complex<double>(-0.25 * (wa * wa), -w * P.tshift));
        tab[RF_FTAB * j + 1] = cq.real();
        tab[RF_FTAB * j + 2] = cq.imag();
    }
}

// tw[2*(l+m)], tw[2*(l+m)+1] = exp(i*pi*m/l) for l = 1,2,4,..,n/2 and m < l  (fork.cpp:50-51, signi=+1)
inline void rf_fill_twiddles(double *tw, int nsamp)
{
    tw[0] = tw[1] = 0.0;
    for (int l = 1; l < nsamp; l <<= 1)
        for (int m = 0; m < l; m++) {
            std::complex<double> w = std::exp(std::complex<double>(0.0, M_PI * (double)(1 * m) / (double)l));
            tw[2 * (l + m)] = w.real();
            tw[2 * (l + m) + 1] = w.imag();
        }
}

}  // namespace bh
