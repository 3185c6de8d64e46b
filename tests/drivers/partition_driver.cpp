// partition_csr / row_block_offsets (include/rocalution/distribute.hpp; clients/include/common.hpp:56-431 of the reference):
// pure host arithmetic, checked here for 1, 2, 3, 5 and 7 ranks on a symmetric 5-point grid matrix and on a random
// unsymmetric pattern (values are small integers: every sum is exact).
//   * the row blocks follow the reference's rule and cover the matrix, the pieces hold all its entries
//   * what rank r sends to q is what q expects from r (same rows, same order); neighbourhood is symmetric
//   * the distributed product (interior * local x + ghost * received x, the exchange simulated from the send lists) equals A x
//   * symmetric pattern: the send list is the one the reference builds from the rank's OWN rows
#include <rocalution/rocalution.hpp>

#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>

using namespace rocalution;

#define REQUIRE(cond)                                                             \
    do                                                                            \
    {                                                                             \
        if(!(cond))                                                               \
        {                                                                         \
            std::printf("partition_driver: failed at line %d: %s\n", __LINE__, #cond); \
            return 1;                                                             \
        }                                                                         \
    } while(0)

struct Csr
{
    int64_t              n;
    std::vector<PtrType> rp;
    std::vector<int>     col;
    std::vector<double>  val;
};

static Csr grid5(int nx, int ny)
{
    Csr A;
    A.n = (int64_t)nx * ny;
    A.rp.push_back(0);
    for(int j = 0; j < ny; ++j)
        for(int i = 0; i < nx; ++i)
        {
            const int r = j * nx + i;
            if(j > 0) { A.col.push_back(r - nx); A.val.push_back(-1); }
            if(i > 0) { A.col.push_back(r - 1); A.val.push_back(-2); }
            A.col.push_back(r); A.val.push_back(9);
            if(i < nx - 1) { A.col.push_back(r + 1); A.val.push_back(-3); }
            if(j < ny - 1) { A.col.push_back(r + nx); A.val.push_back(-4); }
            A.rp.push_back((PtrType)A.col.size());
        }
    return A;
}

static unsigned lcg(unsigned& s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

static Csr random_pattern(int n, unsigned seed)
{
    Csr A;
    A.n = n;
    A.rp.push_back(0);
    for(int r = 0; r < n; ++r)
    {
        std::set<int> cols;
        cols.insert(r);
        const int k = (int)(lcg(seed) % 6);
        for(int e = 0; e < k; ++e)
            cols.insert((int)(lcg(seed) % (unsigned)n));
        for(std::set<int>::const_iterator it = cols.begin(); it != cols.end(); ++it)
        {
            A.col.push_back(*it);
            A.val.push_back((double)((int)(lcg(seed) % 7) - 3));
        }
        A.rp.push_back((PtrType)A.col.size());
    }
    return A;
}

static int check(const Csr& A, int ranks, bool symmetric)
{
    const std::vector<int64_t> off = row_block_offsets(A.n, ranks);
    REQUIRE(off.front() == 0 && off.back() == A.n);
    for(int r = 0; r < ranks; ++r)
        REQUIRE(off[r + 1] - off[r] == A.n / ranks + (r < A.n % ranks ? 1 : 0));
    std::vector<RankPiece<double>> P;
    for(int r = 0; r < ranks; ++r)
        P.push_back(partition_csr<double>(r, ranks, A.n, A.rp.data(), A.col.data(), A.val.data()));
    size_t entries = 0;
    for(int r = 0; r < ranks; ++r)
    {
        REQUIRE(P[r].local_nrow == off[r + 1] - off[r]);
        entries += P[r].int_col.size() + P[r].gst_col.size();
        REQUIRE(P[r].recv_offset.size() == P[r].peers.size() + 1 && P[r].send_offset.size() == P[r].peers.size() + 1);
        REQUIRE((size_t)P[r].send_offset.back() == P[r].boundary.size());
        REQUIRE((size_t)P[r].recv_offset.back() == P[r].recv_global.size());
    }
    REQUIRE(entries == A.col.size());
    // send list of r for q == receive list of q from r
    for(int r = 0; r < ranks; ++r)
        for(size_t k = 0; k < P[r].peers.size(); ++k)
        {
            const int q = P[r].peers[k];
            size_t    kk = 0;
            while(kk < P[q].peers.size() && P[q].peers[kk] != r)
                ++kk;
            REQUIRE(kk < P[q].peers.size()); // symmetric neighbourhood
            const int ns = P[r].send_offset[k + 1] - P[r].send_offset[k];
            REQUIRE(ns == P[q].recv_offset[kk + 1] - P[q].recv_offset[kk]);
            for(int e = 0; e < ns; ++e)
                REQUIRE(P[r].boundary[(size_t)P[r].send_offset[k] + e] + off[r] == P[q].recv_global[(size_t)P[q].recv_offset[kk] + e]);
            if(symmetric)
            {
                // the reference's construction: r's own rows with a column owned by q, ascending, once each
                std::vector<int> own;
                for(int64_t i = off[r]; i < off[r + 1]; ++i)
                    for(PtrType j = A.rp[i]; j < A.rp[i + 1]; ++j)
                        if(A.col[j] >= off[q] && A.col[j] < off[q + 1])
                        {
                            if(own.empty() || own.back() != (int)(i - off[r]))
                                own.push_back((int)(i - off[r]));
                            break;
                        }
                REQUIRE((int)own.size() == ns);
                for(int e = 0; e < ns; ++e)
                    REQUIRE(own[e] == P[r].boundary[(size_t)P[r].send_offset[k] + e]);
            }
        }
    // distributed product
    std::vector<double> x((size_t)A.n), y((size_t)A.n, 0.0);
    unsigned            seed = 12345u + (unsigned)ranks;
    for(int64_t i = 0; i < A.n; ++i)
        x[i] = (double)((int)(lcg(seed) % 11) - 5);
    for(int64_t i = 0; i < A.n; ++i)
        for(PtrType j = A.rp[i]; j < A.rp[i + 1]; ++j)
            y[i] += A.val[j] * x[A.col[j]];
    for(int r = 0; r < ranks; ++r)
    {
        std::vector<double> recv(P[r].recv_global.size());
        for(size_t k = 0; k < P[r].peers.size(); ++k)
        {
            const int q = P[r].peers[k];
            size_t    kk = 0;
            while(P[q].peers[kk] != r)
                ++kk;
            for(int e = 0; e < P[r].recv_offset[k + 1] - P[r].recv_offset[k]; ++e)
                recv[(size_t)P[r].recv_offset[k] + e] = x[(size_t)(off[q] + P[q].boundary[(size_t)P[q].send_offset[kk] + e])];
        }
        for(int64_t i = 0; i < P[r].local_nrow; ++i)
        {
            double s = 0.0;
            for(PtrType j = P[r].int_rp[i]; j < P[r].int_rp[i + 1]; ++j)
                s += P[r].int_val[j] * x[(size_t)(off[r] + P[r].int_col[j])];
            for(PtrType j = P[r].gst_rp[i]; j < P[r].gst_rp[i + 1]; ++j)
                s += P[r].gst_val[j] * recv[(size_t)P[r].gst_col[j]];
            REQUIRE(s == y[(size_t)(off[r] + i)]);
        }
    }
    return 0;
}

static void dump_list(const char* name, const std::vector<int>& v)
{
    std::printf(" %s", name);
    for(size_t i = 0; i < v.size(); ++i)
        std::printf("%c%d", i ? ',' : '=', v[i]);
    if(v.empty())
        std::printf("=");
}

// "dump <ranks>": the pieces of the 7 x 9 grid matrix, one line per rank (compared with rocalution_amd/distributed.py, the
// implementation the 2-rank runs validate, in tests/test_cpu_host.py)
static int dump(int ranks)
{
    const Csr A = grid5(7, 9);
    for(int r = 0; r < ranks; ++r)
    {
        const RankPiece<double> P = partition_csr<double>(r, ranks, A.n, A.rp.data(), A.col.data(), A.val.data());
        std::printf("rank=%d", r);
        dump_list("peers", P.peers);
        dump_list("recv_offset", P.recv_offset);
        dump_list("send_offset", P.send_offset);
        dump_list("boundary", P.boundary);
        dump_list("int_rp", std::vector<int>(P.int_rp.begin(), P.int_rp.end()));
        dump_list("int_col", P.int_col);
        dump_list("gst_rp", std::vector<int>(P.gst_rp.begin(), P.gst_rp.end()));
        dump_list("gst_col", P.gst_col);
        dump_list("int_val", std::vector<int>(P.int_val.begin(), P.int_val.end()));
        dump_list("gst_val", std::vector<int>(P.gst_val.begin(), P.gst_val.end()));
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char** argv)
{
    if(argc == 3 && std::string(argv[1]) == "dump")
        return dump(std::atoi(argv[2]));
    const Csr grid = grid5(7, 9), rnd = random_pattern(53, 7u), tiny = random_pattern(2, 3u);
    const int ranks[] = {1, 2, 3, 5, 7};
    for(int k = 0; k < 5; ++k)
    {
        if(check(grid, ranks[k], true))
            return 1;
        if(check(rnd, ranks[k], false))
            return 1;
        if(check(tiny, ranks[k], false)) // more ranks than rows: empty blocks
            return 1;
    }
    std::printf("partition_driver ok\n");
    return 0;
}
