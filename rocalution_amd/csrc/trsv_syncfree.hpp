// trsv_syncfree.hpp -- sync-free grouped triangular solve (trsv_syncfree.hip: k_trsv_sf), used by trisolve.hip, whose analysis
// (build_sf_plan) decides for the form, sorts the positions by group level and cuts them into units.
#pragma once

#include "common.hpp"

namespace ramd
{

constexpr int kGrpMax = 8; // rows of a row group (a supernode run is cut into groups of at most this many rows)
constexpr int kSfKW   = 6; // out-of-group entries per lane of a row, at most

// the plan arrays beyond order / pos / diag / w of the TriPlan that owns it
struct SfPlan
{
    int   nunits = 0, lpr = 0, maxm = 1, wout = 0, ngroups = 0, nglev = 0;
    // [4 nunits] {first position,
    //             rows | entries per lane << 8 | lanes per row in use << 12 | one group only << 16 | rows of the longest group << 17,
    //             first entry of the unit in ecol / eval, last dependency (position)}
    int*  uinfo  = nullptr;
    int*  punit  = nullptr; // [n] unit of a position
    int*  ufar   = nullptr; // [nunits] a position a few levels back of the unit's dependencies (k_sf_far)
    int*  pinfo  = nullptr; // [n] per position: row number inside its group | rows of the group << 4 | out-of-group entries << 8
    // out-of-group entries, per unit a block of kw planes of rows x nl lanes (rows = rows of the unit, nl = lanes per row in use:
    // only the lanes that hold entries are stored): entry e of a row sits in lane e / kw of the row, plane e % kw
    int*  ecol   = nullptr; // positions the entries refer to (-1: no entry)
    void* eval   = nullptr;
    void* gcoef  = nullptr; // [8 n] in-group coefficients of a position (groups of more than one row)
    // [n], fp64 plans: the reciprocal iterate of the diagonal that the division sequence of gfx950 forms from the divisor alone
    // (sf_recip_iterate), 0 where the divisor is outside the range in which that sequence scales nothing
    void* rdiag  = nullptr;
    unsigned* tickets = nullptr; // ticket words of a launch (start tickets + one word per ticket stream), zeroed before every launch
    bool  infirst = false;  // the in-group entries come first in the order of the host loop (upper solve)
    int64_t nentries = 0;   // slots of ecol / eval
};
void sf_release(SfPlan** sp);

// slots the unit {rows, kw, lanes per row in use} takes in ecol / eval
inline int64_t sf_unit_slots(int rows, int kw, int nl)
{
    return (int64_t)rows * kw * nl;
}

// fills ecol / eval / gcoef / diag / rdiag / punit and the last-dependency word of every unit from the CSR arrays and the
// position order (uinfo words 0-2 and pinfo written by the caller); *nodiag_out: a row without a stored diagonal was met
template <typename T>
int sf_fill(SfPlan* S, int n, bool lower, bool reverse, const int* order, const int* pos, const int* rp, const int* ci, const T* val,
            T* diag, bool* nodiag_out);

// one launch: w = sentinel, then the persistent waves.  dm 0: unit diagonal, 1: divide by diag, 2: multiply by diag.
// out (may be null): the result at order[p] as well
template <typename T>
int sf_run(const SfPlan* S, int n, int dm, const T* diag, T* w, const int* order, const T* rhs_src, const int* rhs_idx, T* out);

} // namespace ramd
