// Host-only exercise of LocalMatrix pieces that need no accelerator: COO input (stable row sort), Check(),
// CopyToCOO / LeaveDataPtrCOO, UpdateValuesCSR on host storage.
#include <rocalution/rocalution.hpp>

#include <cstdio>
#include <iostream>

using namespace rocalution;

int main()
{
    // 4x5, unsorted COO with two entries in row 2 given in "wrong" column order and an empty row 1
    const int    nnz   = 6;
    int*         row   = new int[nnz]{2, 0, 3, 2, 0, 3};
    int*         col   = new int[nnz]{4, 1, 0, 1, 0, 3};
    double*      val   = new double[nnz]{24., 1., 30., 21., 0.5, 33.};
    LocalMatrix<double> A;
    A.SetDataPtrCOO(&row, &col, &val, "A", nnz, 4, 5);
    if(row != NULL || col != NULL || val != NULL)
        return 10; // ownership is taken, caller pointers are nulled (local_matrix.cpp:782-850)
    if(A.GetM() != 4 || A.GetN() != 5 || A.GetNnz() != 6)
        return 11;
    int    rp[5], ci[6];
    double va[6];
    A.CopyToCSR(rp, ci, va);
    const int    erp[5] = {0, 2, 2, 4, 6}, eci[6] = {1, 0, 4, 1, 0, 3};
    const double eva[6] = {1., 0.5, 24., 21., 30., 33.};
    for(int i = 0; i < 5; ++i)
        if(rp[i] != erp[i])
            return 12;
    for(int i = 0; i < 6; ++i)
        if(ci[i] != eci[i] || va[i] != eva[i])
            return 13; // storage order inside a row is kept (stable)
    if(!A.Check())
        return 14; // unsorted columns are a warning, not an error
    int *r2 = NULL, *c2 = NULL;
    double* v2 = NULL;
    A.LeaveDataPtrCOO(&r2, &c2, &v2);
    const int er[6] = {0, 0, 2, 2, 3, 3};
    for(int i = 0; i < 6; ++i)
        if(r2[i] != er[i] || c2[i] != eci[i] || v2[i] != eva[i])
            return 15;
    if(A.GetNnz() != 0)
        return 16;
    // Check() catches what the reference's catches
    LocalMatrix<double> B;
    B.AllocateCSR("B", 3, 2, 2);
    int    brp[3] = {0, 2, 3}, bci[3] = {1, 1, 0};
    double bva[3] = {1., 2., 3.};
    B.CopyFromCSR(brp, bci, bva);
    if(B.Check())
        return 17; // duplicated column entry
    bci[1] = 0;
    bva[2] = std::numeric_limits<double>::infinity();
    B.CopyFromCSR(brp, bci, bva);
    if(B.Check())
        return 18; // infinite value
    bva[2] = 3.;
    B.CopyFromCSR(brp, bci, bva);
    if(!B.Check())
        return 19;
    double nv[3] = {7., 8., 9.};
    B.UpdateValuesCSR(nv);
    B.CopyToCSR(brp, bci, bva);
    if(bva[0] != 7. || bva[2] != 9.)
        return 20;
    free_host(&r2);
    free_host(&c2);
    free_host(&v2);
    std::cout << "api_driver ok" << std::endl;
    return 0;
}
