// Host-only exercise of the file IO layer (include/rocalution/io.hpp) -- no accelerator needed:
// read the files the genuine rocALUTION library wrote (tests/golden/io) and write them again.
#include <rocalution/rocalution.hpp>

#include <iostream>
#include <string>

using namespace rocalution;

int main(int argc, char** argv)
{
    if(argc < 3)
        return 2;
    const std::string in = argv[1], out = argv[2];
    {
        LocalMatrix<double> A;
        A.ReadFileMTX(in + "/ref_A.mtx");
        A.WriteFileMTX(out + "/A_from_mtx.mtx");
        A.WriteFileCSR(out + "/A_from_mtx.csr");
        LocalMatrix<double> B;
        B.ReadFileCSR(in + "/ref_A.csr");
        B.WriteFileMTX(out + "/A_from_csr.mtx");
        B.WriteFileCSR(out + "/A_from_csr.csr");
        LocalMatrix<float> F; // values are stored as double in the file whatever the precision
        F.ReadFileCSR(in + "/ref_A.csr");
        F.WriteFileCSR(out + "/A_float.csr");
    }
    for(const char* nm : {"sym", "pat", "gen"})
    {
        LocalMatrix<double> M;
        M.ReadFileMTX(in + "/in_" + nm + ".mtx");
        M.WriteFileCSR(out + "/read_" + nm + ".csr");
    }
    {
        LocalVector<double> x;
        x.ReadFileASCII(in + "/ref_x.dat");
        x.WriteFileASCII(out + "/x_from_ascii.dat");
        LocalVector<double> y;
        y.ReadFileBinary(in + "/ref_x.bin");
        y.WriteFileBinary(out + "/x_from_bin.bin");
        y.WriteFileASCII(out + "/x_from_bin.dat");
        LocalVector<float> f;
        f.ReadFileBinary(in + "/ref_x.bin");
        f.WriteFileBinary(out + "/x_float.bin");
        LocalVector<int> p;
        p.Allocate("p", 5);
        int pv[5] = {4, 0, 3, 1, 2};
        p.CopyFromData(pv);
        p.WriteFileBinary(out + "/perm.bin");
        LocalVector<int> q;
        q.ReadFileBinary(out + "/perm.bin");
        int qv[5];
        q.CopyToData(qv);
        for(int i = 0; i < 5; ++i)
            if(qv[i] != pv[i])
                return 3;
    }
    std::cout << "io_driver ok" << std::endl;
    return 0;
}
