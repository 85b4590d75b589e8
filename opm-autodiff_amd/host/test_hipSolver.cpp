// Counterpart of the reference's tests/test_cusparseSolver.cpp:49-131 for the HIP backend: read matr33.txt / rhs3.txt
// (MatrixMarket, ISTL_STRUCT blocked 3 3), run bda::hipSolverBackend<3>::solve_system + get_result with tol / maxit
// from the command line, print the solution.  The expected vector is checked by the calling pytest
// (tests/test_gpu_host_cpp.py) against the fixture in tests/golden/linalg/expected.json.
//   usage: test_hipSolver matr33.txt rhs3.txt tol maxit reorder [wells|mswells|msonly|-] [linsolver]     (linsolver: ilu0 | cpr | cpr_trueimpes | cpr_quasiimpes;
//   a CPR run solves twice, the second time behind recreateCprHierarchy(): the --cpr-reuse-setup=1 path of the plug-in)
//   wells: one standard well; mswells: that standard well and one multisegment well (two segments, three perforations); msonly: the
//   multisegment well alone - WellContributions::getNumWells() then counts wells the C arrays do not hold (bda/WellContributions.hpp:164-166)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <map>
#include <sstream>

#include "hipSolverBackend.hpp"

static bool readBlocked(const std::string& file, int& Nb, std::vector<int>& rows, std::vector<int>& cols, std::vector<double>& vals) {
    std::ifstream f(file);
    if (!f) return false;
    std::string line;
    int bs = 1;
    while (std::getline(f, line) && !line.empty() && line[0] == '%') {
        if (line.find("ISTL_STRUCT") != std::string::npos) { std::istringstream is(line); std::string a, b, c; is >> a >> b >> c >> bs; }
    }
    int n, m, nnz;
    std::istringstream(line) >> n >> m >> nnz;
    std::map<std::pair<int, int>, std::vector<double>> blocks;
    for (int e = 0; e < nnz; ++e) {
        int r, c; double v;
        f >> r >> c >> v; --r; --c;
        auto& b = blocks[{r / bs, c / bs}];
        if (b.empty()) b.assign(bs * bs, 0.0);
        b[(r % bs) * bs + c % bs] = v;
    }
    Nb = n / bs;
    rows.assign(Nb + 1, 0);
    for (auto& kv : blocks) rows[kv.first.first + 1]++;
    for (int i = 0; i < Nb; ++i) rows[i + 1] += rows[i];
    for (auto& kv : blocks) { cols.push_back(kv.first.second); vals.insert(vals.end(), kv.second.begin(), kv.second.end()); }
    return bs == 3;
}
static bool readVector(const std::string& file, std::vector<double>& b) {
    std::ifstream f(file);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line) && !line.empty() && line[0] == '%') {}
    int n, one;
    std::istringstream(line) >> n >> one;
    b.resize(n);
    for (auto& v : b) f >> v;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s matrix rhs tol maxit reorder\n", argv[0]); return 2; }
    int Nb;
    std::vector<int> rows, cols;
    std::vector<double> vals, rhs;
    if (!readBlocked(argv[1], Nb, rows, cols, vals)) throw std::runtime_error("Could not read matrix file");
    if (!readVector(argv[2], rhs)) throw std::runtime_error("Could not read rhs file");
    const double tolerance = std::atof(argv[3]);
    const int maxit = std::atoi(argv[4]);
    std::unique_ptr<bda::hipSolverBackend<3>> backend;
    try {
        backend.reset(new bda::hipSolverBackend<3>(/*verbosity=*/0, maxit, tolerance, /*deviceID=*/0, argv[5], /*w=*/1.0, argc > 7 ? argv[7] : "ilu0",
                                                   /*cpr_reuse_setup=*/argc > 7 ? 1 : 3));
    } catch (const std::logic_error& error) {
        std::fprintf(stderr, "Problem with initializing a device: %s\n", error.what());  // the reference skips here
        return 77;
    }
    Opm::WellContributions wellContribs;
    const std::string wellMode = argc > 6 ? argv[6] : "-";
    if (wellMode == "wells" || wellMode == "mswells") {
        // one standard well with two perforations (cells 1 and Nb - 2), filled the way StandardWellEval does
        // (wells/StandardWellEval.cpp:1206-1250: C, D, B); the shim reads the arrays back - directly from the stand-in
        // class, or through getHostArrays when built with OPMHIP_USE_OPM_HEADERS
        wellContribs.setBlockSize(3, 4);
        wellContribs.addNumBlocks(2);
        wellContribs.alloc();
        const int wcols[2] = {1, Nb - 2};
        double C[24], B[24], D[16];
        for (int i = 0; i < 24; ++i) { C[i] = 0.01 * (1 + (i * 7) % 5); B[i] = 0.02 * (1 + (i * 3) % 7); }
        for (int i = 0; i < 16; ++i) D[i] = (i % 5 == 0) ? 0.5 : 0.01 * (i % 3);
        wellContribs.addMatrix(Opm::WellContributions::MatrixType::C, wcols, C, 2);
        wellContribs.addMatrix(Opm::WellContributions::MatrixType::D, nullptr, D, 1);
        wellContribs.addMatrix(Opm::WellContributions::MatrixType::B, wcols, B, 2);
    }
    if (wellMode == "mswells" || wellMode == "msonly") {
        // one multisegment well the way MultisegmentWell::addWellContribution hands it over (wells/MultisegmentWellEval.cpp:1940-1982: B and C blocked
        // CSR with one pattern, D in CSC - WellContributions.hpp:196-213): segment 0 perforates cell 0, segment 1 cells 2 and Nb - 1
        const unsigned dimW = 4, Mb = 2, M = Mb * dimW;
        std::vector<unsigned> Bcols = {0u, 2u, (unsigned)(Nb - 1)}, Brows = {0u, 1u, 3u};
        std::vector<double> Bv(3 * 12), Cv(3 * 12);
        for (int i = 0; i < 36; ++i) { Bv[i] = 0.03 * (1 + (i * 5) % 7) - 0.05; Cv[i] = 0.02 * (1 + (i * 3) % 5); }
        // D (8 x 8): diagonally dominant, both segments coupled; CSC, every entry present
        std::vector<double> Dv;
        std::vector<int> Dcp(M + 1, 0), Dri;
        for (unsigned c = 0; c < M; ++c) {
            for (unsigned r = 0; r < M; ++r) {
                Dv.push_back(r == c ? 2.0 + 0.1 * r : 0.05 * ((r * 3 + c * 5) % 4) - 0.04);
                Dri.push_back((int)r);
            }
            Dcp[c + 1] = (int)Dv.size();
        }
        wellContribs.addMultisegmentWellContribution(3, dimW, Mb, Bv, Bcols, Brows, (unsigned)(Dv.size() / (dimW * dimW)), Dv.data(), Dcp.data(), Dri.data(), Cv);
    }
    bda::BdaResult result;
    std::vector<double> x(rhs.size());
    const bda::SolverStatus st = backend->solve_system(3 * Nb, 9 * (int)cols.size(), 3, vals.data(), rows.data(), cols.data(), rhs.data(), wellContribs, result);
    if (st != bda::SolverStatus::BDA_SOLVER_SUCCESS) return 3;
    if (argc > 7 && std::string(argv[7]) != "ilu0") {   // the first Newton iteration of the next time step: hierarchy anew, same answer
        std::vector<double> v2(vals), b2(rhs);
        backend->recreateCprHierarchy();
        if (backend->solve_system(3 * Nb, 9 * (int)cols.size(), 3, v2.data(), rows.data(), cols.data(), b2.data(), wellContribs, result) != bda::SolverStatus::BDA_SOLVER_SUCCESS) return 3;
    }
    backend->get_result(x.data());
    std::printf("converged %d iterations %d reduction %.17g\n", (int)result.converged, result.iterations, result.reduction);
    for (double v : x) std::printf("%.17g\n", v);
    return 0;
}
