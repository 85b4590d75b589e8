// Drives Newton iterations through Opm::HipLinearizer the way Flow drives its linearizer - model().linearizer().linearizeDomain(),
// .jacobian(), .residual(), solution(0), invalidateAndUpdateIntensiveQuantities (flow/BlackoilModelEbos.hpp:339-340, 424, 526-527,
// 552-562) - against own minimal stand-ins for the Simulator / Problem / Model types the class template names, and checks every
// iteration bit for bit against the plain C-ABI sequence BlackoilModelHip runs on a second context (opmhip_assemble with host
// copies -> opmhip_solve_system -> opmhip_update).  Input: a case file of decks.py:write_case_binary.
//   usage: test_HipLinearizer case.bin host|device dt_seconds iterations state_out.bin
#include <array>
#include <cstdio>
#include <cstdlib>

#include "CaseFile.hpp"
#include "HipLinearizer.hpp"

namespace stub {
// ---- what opm-models / dune-istl provide in a real tree, reduced to what the linearizer touches --------------------------------
struct PrimaryVariables {
    std::array<double, 3> x{};
    unsigned char meaning = 0;
    double& operator[](int i) { return x[i]; }
    double operator[](int i) const { return x[i]; }
    unsigned char primaryVarsMeaning() const { return meaning; }
    void setPrimaryVarsMeaning(unsigned char m) { meaning = m; }
};
using SolutionVector = std::vector<PrimaryVariables>;
using GlobalEqVector = std::vector<std::array<double, 3>>;   // Dune::BlockVector<FieldVector<double, 3>>: contiguous triples
struct RateVector {                                          // value + derivatives per equation (DenseAd evaluations in Flow)
    double v[3] = {0, 0, 0}, d[3][3] = {{0}};
    double value(int e) const { return v[e]; }
    double derivative(int e, int var) const { return d[e][var]; }
};
struct Simulator;
// Linear::IstlSparseMatrixAdapter: owns a BCRSMatrix<MatrixBlock<double, 3, 3>> built from a sparsity pattern
class SparseMatrixAdapter {
public:
    struct Block { double m[3][3]; double* operator[](int r) { return m[r]; } };
    struct Row {
        Block* b; const int* c; int n;
        Block& operator[](int col) { for (int k = 0; k < n; ++k) if (c[k] == col) return b[k]; throw std::out_of_range("no such block"); }
    };
    struct Bcrs {
        std::vector<int> rowptr, col;
        std::vector<Block> val;
        Row operator[](int i) { return Row{&val[rowptr[i]], &col[rowptr[i]], rowptr[i + 1] - rowptr[i]}; }
    };
    explicit SparseMatrixAdapter(const Simulator&) {}
    void reserve(const std::vector<std::set<unsigned>>& pattern) {
        A_.rowptr.assign(1, 0);
        for (const auto& r : pattern) { A_.col.insert(A_.col.end(), r.begin(), r.end()); A_.rowptr.push_back((int)A_.col.size()); }
        A_.val.assign(A_.col.size(), Block{});
    }
    Bcrs& istlMatrix() { return A_; }
    const Bcrs& istlMatrix() const { return A_; }
private:
    Bcrs A_;
};
struct NewtonMethod {
    int it = 0;
    int numIterations() const { return it; }
    void setIterationIndex(int i) { it = i; }
};
struct Model {
    hipcase::Case* C = nullptr;
    int N = 0;
    SolutionVector sol;
    NewtonMethod newton;
    std::vector<std::vector<int>> nb;
    size_t numGridDof() const { return (size_t)N; }
    double dofTotalVolume(int i) const { return (*C)["volume"].f64()[i]; }
    const std::vector<int>& stencilNeighbors(int i) const { return nb[i]; }
    SolutionVector& solution(unsigned) { return sol; }
    const SolutionVector& solution(unsigned) const { return sol; }
    NewtonMethod& newtonMethod() { return newton; }
    const NewtonMethod& newtonMethod() const { return newton; }
};
struct Problem {
    hipcase::Case* C = nullptr;
    opmhip_fluid fl{};
    std::map<std::pair<int, int>, int> entry;   // (i, j) -> position in the case's pattern
    const opmhip_fluid& hipFluidTables() const { return fl; }
    double transmissibility(int i, int j) const { return (*C)["trans"].f64()[entry.at({i, j})]; }
    double faceArea(int i, int j) const { return (*C)["area"].f64()[entry.at({i, j})]; }
    double thresholdPressure(int, int) const { return 0.0; }
    double porosity(int i) const { return (*C)["poro"].f64()[i]; }
    double dofCenterDepth(int i) const { return (*C)["depth"].f64()[i]; }
    int pvtRegionIndex(int) const { return 0; }
    int satnumRegionIndex(int) const { return 0; }
    double maxGasDissolutionFactor(unsigned, int) const { return 1e300; }   // no DRSDT
    void source(RateVector& rate, int i, unsigned) const {
        rate = RateVector();
        if (C->count("source")) for (int e = 0; e < 3; ++e) rate.v[e] = (*C)["source"].f64()[(size_t)i * 3 + e];
    }
#ifdef STUB_SOURCE_DOFS   // the second build of this driver: a problem that names the dofs with sources (the perforated cells)
    mutable std::vector<int> dofs;
    const std::vector<int>& sourceDofs() const {
        if (dofs.empty() && C->count("source")) {
            const int N = (int)(*C)["source"].count / 3;
            for (int i = 0; i < N; ++i)
                for (int e = 0; e < 3; ++e)
                    if ((*C)["source"].f64()[(size_t)i * 3 + e] != 0.0) { dofs.push_back(i); break; }
        }
        return dofs;
    }
#endif
};
struct Simulator {
    Problem problem_;
    Model model_;
    double dt = 86400.0;
    Problem& problem() { return problem_; }
    const Problem& problem() const { return problem_; }
    Model& model() { return model_; }
    const Model& model() const { return model_; }
    double timeStepSize() const { return dt; }
};
struct TypeTag {};
}  // namespace stub

namespace Opm::Properties {
template <> struct Simulator<stub::TypeTag, stub::TypeTag> { using type = stub::Simulator; };
template <> struct SparseMatrixAdapter<stub::TypeTag, stub::TypeTag> { using type = stub::SparseMatrixAdapter; };
template <> struct GlobalEqVector<stub::TypeTag, stub::TypeTag> { using type = stub::GlobalEqVector; };
template <> struct SolutionVector<stub::TypeTag, stub::TypeTag> { using type = stub::SolutionVector; };
template <> struct RateVector<stub::TypeTag, stub::TypeTag> { using type = stub::RateVector; };
template <> struct Linearizer<stub::TypeTag, stub::TypeTag> { using type = Opm::HipLinearizer<stub::TypeTag>; };   // the line a maintainer adds
}  // namespace Opm::Properties

#define CK(ctx, call) do { int rc_ = (call); if (rc_ != OPMHIP_SUCCESS) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, opmhip_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s case.bin host|device dt iterations state_out.bin\n", argv[0]); return 2; }
    auto C = hipcase::read(argv[1]);
    const bool hostCopies = std::string(argv[2]) == "host";
    const double dt = std::atof(argv[3]);
    const int iterations = std::atoi(argv[4]);
    const int N = (int)C["rowptr"].count - 1, nnz = (int)C["col"].count;
    stub::Simulator sim;
    sim.dt = dt;
    sim.problem_.C = sim.model_.C = &C;
    sim.problem_.fl = hipcase::fluid(C);
    sim.model_.N = N;
    sim.model_.nb.resize(N);
    for (int i = 0; i < N; ++i)
        for (int k = C["rowptr"].i32()[i]; k < C["rowptr"].i32()[i + 1]; ++k) {
            const int j = C["col"].i32()[k];
            sim.problem_.entry[{i, j}] = k;
            if (j != i) sim.model_.nb[i].push_back(j);
        }
    sim.model_.sol.resize(N);
    for (int i = 0; i < N; ++i) {
        for (int v = 0; v < 3; ++v) sim.model_.sol[i][v] = C["pv"].f64()[(size_t)i * 3 + v];
        sim.model_.sol[i].meaning = C["meaning"].u8()[i];
    }
    opmhip_config cfg;
    opmhip_default_config(&cfg);
    cfg.reorder = OPMHIP_REORDER_LINE_COLORING;
    using Linearizer = Opm::GetPropType<stub::TypeTag, Opm::Properties::Linearizer>;
    Linearizer lin;
    try { lin.init(sim, &cfg, hostCopies); }
    catch (const std::exception& e) { std::fprintf(stderr, "init: %s\n", e.what()); return 77; }
    if (lin.numRows() != N || lin.numBlocks() != nnz) { std::fprintf(stderr, "pattern from the stencils differs from the case's\n"); return 3; }
    // the yardstick: the plain C-ABI sequence of BlackoilModelHip on a context of its own
    opmhip_ctx* ref = nullptr;
    if (opmhip_create(&cfg, &ref) != OPMHIP_SUCCESS) return 77;
    opmhip_fluid fl = hipcase::fluid(C);
    CK(ref, opmhip_set_pattern(ref, N, nnz, C["rowptr"].i32(), C["col"].i32()));
    CK(ref, opmhip_set_fluid(ref, &fl));
    CK(ref, opmhip_set_static(ref, C["trans"].f64(), C["area"].f64(), nullptr, C["poro"].f64(), C["volume"].f64(), C["depth"].f64(), nullptr, nullptr, nullptr));
    CK(ref, opmhip_set_state(ref, C["pv"].f64(), C["meaning"].u8()));
    if (C.count("source")) CK(ref, opmhip_set_source(ref, C["source"].f64(), nullptr));
    std::vector<double> J((size_t)nnz * 9), r((size_t)N * 3), rdev((size_t)N * 3), pvA((size_t)N * 3), pvB((size_t)N * 3);
    std::vector<unsigned char> mA(N), mB(N);
    for (int it = 0; it < iterations; ++it) {
        sim.model().newtonMethod().setIterationIndex(it);      // assembleReservoir, BlackoilModelEbos.hpp:422
        lin.linearizeDomain();
        CK(ref, opmhip_assemble(ref, dt, it, J.data(), r.data()));
        bool same;
        if (hostCopies) {
            const double* Jl = &(lin.jacobian().istlMatrix()[0][0][0][0]);
            const double* rl = &(lin.residual()[0][0]);
            same = std::memcmp(Jl, J.data(), J.size() * 8) == 0 && std::memcmp(rl, r.data(), r.size() * 8) == 0;
        } else {
            CK(lin.context(), opmhip_get_rhs(lin.context(), rdev.data()));
            same = std::memcmp(rdev.data(), r.data(), r.size() * 8) == 0;
        }
        opmhip_result ra{}, rb{};
        CK(lin.context(), opmhip_solve_system(lin.context(), N * 3, nnz * 9, 3, nullptr, nullptr, nullptr, nullptr, nullptr, &ra));
        CK(ref, opmhip_solve_system(ref, N * 3, nnz * 9, 3, nullptr, nullptr, nullptr, nullptr, nullptr, &rb));
        lin.updateSolutionOnDevice(1.0);
        CK(ref, opmhip_update(ref, nullptr, 1.0, nullptr));
        CK(ref, opmhip_get_state(ref, pvB.data(), mB.data()));
        bool sameState = true;
        for (int i = 0; i < N && sameState; ++i) {
            const auto& q = sim.model().solution(0)[i];
            sameState = q.meaning == mB[i] && q[0] == pvB[(size_t)i * 3] && q[1] == pvB[(size_t)i * 3 + 1] && q[2] == pvB[(size_t)i * 3 + 2];
        }
        std::printf("iteration %d linear %d / %d system %s state %s\n", it, ra.iterations, rb.iterations, same ? "identical" : "DIFFERS", sameState ? "identical" : "DIFFERS");
        if (!same || !sameState || ra.iterations != rb.iterations) return 5;
        // the host changes solution(0) (as a host-side Newton update would) and tells the model: invalidateAndUpdateIntensiveQuantities
        for (int i = 0; i < N; i += 7) {
            auto& q = sim.model().solution(0)[i];
            q[0] += 1e-4;
            pvB[(size_t)i * 3] = q[0];
        }
        lin.invalidateAndUpdateIntensiveQuantities(0);
        CK(ref, opmhip_set_state(ref, pvB.data(), mB.data()));
    }
    lin.solutionToHost();
    for (int i = 0; i < N; ++i) {
        const auto& q = sim.model().solution(0)[i];
        for (int v = 0; v < 3; ++v) pvA[(size_t)i * 3 + v] = q[v];
        mA[i] = q.meaning;
    }
    std::ofstream o(argv[5], std::ios::binary);
    o.write(reinterpret_cast<const char*>(pvA.data()), (std::streamsize)(pvA.size() * 8));
    o.write(reinterpret_cast<const char*>(mA.data()), (std::streamsize)mA.size());
    opmhip_destroy(ref);
    std::printf("ok\n");
    return 0;
}
