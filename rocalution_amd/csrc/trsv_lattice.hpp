// trsv_lattice.hpp -- register/LDS-resident triangular solve for lattice operators (trsv_lattice.hip), used by trisolve.hip.
#pragma once

#include "common.hpp"

namespace ramd
{

struct LatPlan; // opaque: coefficients in sweep order, face buffers, ticket counter

struct LatInfo
{
    int    nx, ny, nz; // lattice the triangle was recognised on
    int    npencil, nsteps; // 8 x 8 pencils along x; steps a pencil takes (incl. the skew and the padding)
    size_t coef_bytes, face_bytes;
};

// RAMD_ERR_UNSUPPORTED: the triangle is not the lower / upper part of a 5- / 7-point lattice operator in lexicographic
// numbering (or the form is switched off / not worth it at this size): the caller builds one of the general plans.
// unit: the solve leaves the diagonal out (LUSolve's L stage, L/USolve with diag_unit); otherwise it divides by it.
template <typename T>
int lat_build(const ramd_mat_s* m, bool lower, bool unit, LatPlan** out);
// out[r] = solution; in and out may be the same vector
template <typename T>
int  lat_run(LatPlan* P, const T* in, T* out);
void lat_release(LatPlan** P);
bool lat_is_unit(const LatPlan* P);
void lat_info(const LatPlan* P, LatInfo* info);

} // namespace ramd
