// Device-resident model state of an ensemble and the per-step work buffers (model.hip owns the memory).
#pragma once
#include "device_tables.hpp"

namespace spd {

// Raw device pointers handed to the dynamics kernels.  Member-major layouts, see dynamics.hip.
struct ModelPtrs {
    // prognostic spectral state (complex as interleaved doubles)
    double *vor, *div, *t, *tr;  // [M][2][8][992]
    double *ps;                  // [M][2][992]
    double *phi;                 // [M][8][992]   geopotential of the current step (the registry's "phi")
    double *phi_next;            // [M][8][992]   written by spectral_step_kernel for the next step (nullptr: not wanted)
    double *phis;                // [M][992]
    double *tcorh, *qcorh;       // [M][992]   horizontal parts of the orographic diffusion corrections
    // spectral work of the grid <-> spectral export routines (ucos | vcos in the [M][2][8] layout of vor / div)
    double *sv;
    // grid fields of the dynamics (time level j2)
    double *vorg, *divg, *tg2, *trg2, *ug2, *vg2;  // [M][8][NG]
    double *px, *py;                               // [M][NG]
    // tendencies (dynamics writes, physics accumulates) and the other forward-transform inputs
    double *utend, *vtend, *ttend, *trtend, *keg, *utg, *vtg, *uqg, *vqg;  // [M][8][NG]
    double *psdtg;                                                          // [M][NG]
    // forward-transform outputs
    double *specu, *specv;                 // [3][M][8][992]: (utend,vtend), (-uT',-vT'), (-uq,-vq)
    double *spec_tt, *spec_tr, *spec_ke;   // [M][8][992]
    double *spec_ps;                       // [M][992]
};

struct DynDeviceTables {
    const double *dmp, *dmpd, *dmps, *dmp1, *dmp1d, *dmp1s, *elz;  // (31,32)
    const double *xj, *xc, *xd;                                    // (8,8,64), (8,8), (8,8)
    const double *coriol;                                          // 48
    double tcorv[8], qcorv[8], tref[8], tref2[8], tref3[8], dhsx[8], xgeop1[8], xgeop2[8], geo_corf[8];
    double dhs[8], dhsr[8], fsgr[8];
};

}  // namespace spd
