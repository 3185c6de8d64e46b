// ParallelManager::WriteFileASCII / ReadFileASCII (src/base/parallel_manager.cpp:441-743) on host data only: three ranks of a
// 1-D chain write their pattern files, fresh managers read them back; rank 1 also reads a file in the older
// #GLOBAL_SIZE / #LOCAL_SIZE dialect the reference still accepts.  Usage: pm_io_driver <directory>
#include <rocalution/rocalution.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

using namespace rocalution;

static int no_exchange(void*, int, const int*, const void*, const int64_t*, void*, const int64_t*)
{
    return 0;
}
static int no_allreduce(void*, double*, int)
{
    return 0;
}

#define REQUIRE(cond)                                                     \
    do                                                                    \
    {                                                                     \
        if(!(cond))                                                       \
        {                                                                 \
            std::printf("pm_io_driver: failed at line %d: %s\n", __LINE__, #cond); \
            return 1;                                                     \
        }                                                                 \
    } while(0)

struct Pattern
{
    std::vector<int> peers, offsets, boundary;
    int64_t          local;
};

static Pattern pattern_of(int rank)
{
    // ranks 0 | 1 | 2 own 40 | 30 | 30 rows; neighbours exchange two boundary values each way
    Pattern p;
    p.local = rank == 0 ? 40 : 30;
    if(rank == 0)
    {
        p.peers    = {1};
        p.offsets  = {0, 2};
        p.boundary = {38, 39};
    }
    else if(rank == 1)
    {
        p.peers    = {0, 2};
        p.offsets  = {0, 2, 4};
        p.boundary = {0, 1, 28, 29};
    }
    else
    {
        p.peers    = {1};
        p.offsets  = {0, 2};
        p.boundary = {0, 1};
    }
    return p;
}

int main(int argc, char** argv)
{
    REQUIRE(argc == 2);
    const std::string base = std::string(argv[1]) + "/pattern.pm";
    for(int rank = 0; rank < 3; ++rank)
    {
        ramd_comm_t comm = NULL;
        REQUIRE(ramd_comm_init_callback(rank, 3, no_exchange, no_allreduce, NULL, &comm) == RAMD_OK);
        const Pattern   p = pattern_of(rank);
        ParallelManager pm;
        pm.SetMPICommunicator(comm);
        pm.SetGlobalNrow(100);
        pm.SetGlobalNcol(100);
        pm.SetLocalNrow(p.local);
        pm.SetLocalNcol(p.local);
        pm.SetBoundaryIndex((int)p.boundary.size(), p.boundary.data());
        pm.SetReceivers((int)p.peers.size(), p.peers.data(), p.offsets.data());
        pm.SetSenders((int)p.peers.size(), p.peers.data(), p.offsets.data());
        REQUIRE(pm.Status());
        pm.WriteFileASCII(base);

        ParallelManager back;
        back.SetMPICommunicator(comm);
        back.ReadFileASCII(base);
        REQUIRE(back.Status());
        REQUIRE(back.GetGlobalNrow() == 100 && back.GetGlobalNcol() == 100);
        REQUIRE(back.GetLocalNrow() == p.local && back.GetLocalNcol() == p.local);
        REQUIRE(back.GetBoundarySize() == (int)p.boundary.size());
        for(size_t i = 0; i < p.boundary.size(); ++i)
            REQUIRE(back.GetBoundaryIndex()[i] == p.boundary[i]);
        REQUIRE(back.peers() == p.peers);
        REQUIRE(back.send_offset().size() == p.offsets.size() && back.recv_offset().size() == p.offsets.size());
        for(size_t i = 0; i < p.offsets.size(); ++i)
            REQUIRE(back.send_offset()[i] == p.offsets[i] && back.recv_offset()[i] == p.offsets[i]);
        REQUIRE(back.GetNumSenders() == p.offsets.back() && back.GetNumReceivers() == p.offsets.back());

        if(rank == 1)
        {
            // the older dialect: one size for rows and columns
            const std::string old = std::string(argv[1]) + "/old.pm";
            {
                std::ofstream head(old.c_str());
                head << "old.pm.rank.0\nold.pm.rank.1\nold.pm.rank.2\n";
                std::ofstream f((old + ".rank.1").c_str());
                f << "#RANK\n1\n#GLOBAL_SIZE\n100\n#LOCAL_SIZE\n30\n#BOUNDARY_SIZE\n4\n#NUMBER_OF_RECEIVERS\n2\n"
                     "#NUMBER_OF_SENDERS\n2\n#RECEIVERS_RANK\n0\n2\n#SENDERS_RANK\n0\n2\n#RECEIVERS_INDEX_OFFSET\n0\n2\n4\n"
                     "#SENDERS_INDEX_OFFSET\n0\n2\n4\n#BOUNDARY_INDEX\n0\n1\n28\n29\n";
            }
            ParallelManager older;
            older.SetMPICommunicator(comm);
            older.ReadFileASCII(old);
            REQUIRE(older.GetGlobalNrow() == 100 && older.GetGlobalNcol() == 100 && older.GetLocalNrow() == 30
                    && older.GetLocalNcol() == 30 && older.GetBoundarySize() == 4 && older.peers() == p.peers);
            // a pattern of an unsymmetric matrix, as the reference writes it: this rank receives from rank 0 only but
            // sends to ranks 0 and 2 -- read as the union of the two lists with an empty receive piece for rank 2
            const std::string uns = std::string(argv[1]) + "/uns.pm";
            {
                std::ofstream head(uns.c_str());
                head << "uns.pm.rank.0\nuns.pm.rank.1\nuns.pm.rank.2\n";
                std::ofstream f((uns + ".rank.1").c_str());
                f << "#RANK\n1\n#GLOBAL_NROW\n100\n#GLOBAL_NCOL\n100\n#LOCAL_NROW\n30\n#LOCAL_NCOL\n30\n#BOUNDARY_SIZE\n4\n"
                     "#NUMBER_OF_RECEIVERS\n1\n#NUMBER_OF_SENDERS\n2\n#RECEIVERS_RANK\n0\n#SENDERS_RANK\n0\n2\n"
                     "#RECEIVERS_INDEX_OFFSET\n0\n3\n#SENDERS_INDEX_OFFSET\n0\n2\n4\n#BOUNDARY_INDEX\n0\n1\n28\n29\n";
            }
            ParallelManager unsym;
            unsym.SetMPICommunicator(comm);
            unsym.ReadFileASCII(uns);
            REQUIRE(unsym.Status() && unsym.peers() == p.peers);
            REQUIRE(unsym.recv_offset().size() == 3 && unsym.recv_offset()[0] == 0 && unsym.recv_offset()[1] == 3
                    && unsym.recv_offset()[2] == 3);
            REQUIRE(unsym.send_offset().size() == 3 && unsym.send_offset()[1] == 2 && unsym.send_offset()[2] == 4);
            REQUIRE(unsym.GetNumReceivers() == 3 && unsym.GetNumSenders() == 4);
        }
        REQUIRE(ramd_comm_destroy(comm) == RAMD_OK);
    }
    std::printf("pm_io_driver ok\n");
    return 0;
}
