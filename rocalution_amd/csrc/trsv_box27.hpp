// trsv_box27.hpp -- sparse triangular solve on the reference's own 3-D operator (the 27-point stencil of gen_3d_laplacian and
// the ILU(0) factors on its pattern): pencils of the lattice marched along x (trsv_box27.hip), used by trisolve.hip.
#pragma once

#include "common.hpp"

namespace ramd
{

struct BoxPlan; // opaque: coefficients in sweep order, pencil table, outflow records (the hand-off medium), ticket counter, scratch vector

struct BoxInfo
{
    int    nx, ny, nz; // lattice the triangle was recognised on
    int    ntiles, nsteps; // 8 x 8 pencils, steps a pencil takes (incl. the skew)
    size_t coef_bytes;
};

// RAMD_ERR_UNSUPPORTED: the triangle is not the lower / upper part of the full 27-point stencil on an nx x ny x nz lattice in
// lexicographic numbering (or the form is switched off / not worth it at this size): the caller builds one of the general plans.
// unit: the solve leaves the diagonal out (LUSolve's L stage, L / USolve with diag_unit); otherwise it divides by it.
template <typename T>
int box_build(const ramd_mat_s* m, bool lower, bool unit, BoxPlan** out);
// out[r] = solution; in and out must be different vectors (the right-hand side is read ahead of the sweep, the solution written
// behind it, 64 bytes of a grid line at a time)
template <typename T>
int  box_run(BoxPlan* P, const T* in, T* out);
// a vector of the plan's own (n elements): where LUSolve keeps the result of its first stage
void* box_scratch(BoxPlan* P);
void  box_release(BoxPlan** P);
bool  box_is_unit(const BoxPlan* P);
void  box_info(const BoxPlan* P, BoxInfo* info);

} // namespace ramd
