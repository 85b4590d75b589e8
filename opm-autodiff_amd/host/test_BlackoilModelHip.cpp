// Drives time steps of a synthetic black-oil case through the C++ host mirror (Opm::BlackoilModelHip), i.e. the loop
// NonlinearSolverEbos::step -> BlackoilModelEbos::nonlinearIteration would run inside Flow, with every array resident
// on the GPU.  Input: a case file written by opm-autodiff_amd/decks.py:write_case_binary.  Output: per time step the
// Newton and linear iteration counts and, at the end, the state (for the parity test in tests/test_gpu_host_cpp.py).
//   usage: test_BlackoilModelHip case.bin reorder dt_seconds nsteps state_out.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>

#include "BlackoilModelHip.hpp"

struct Arr { int dtype; std::vector<char> bytes; size_t count;
    const int* i32() const { return reinterpret_cast<const int*>(bytes.data()); }
    const double* f64() const { return reinterpret_cast<const double*>(bytes.data()); }
    const unsigned char* u8() const { return reinterpret_cast<const unsigned char*>(bytes.data()); } };

static std::map<std::string, Arr> readCase(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open case file");
    char magic[12];
    f.read(magic, 12);
    if (std::strncmp(magic, "OPMHIPCASE1", 11) != 0) throw std::runtime_error("bad case file");
    std::map<std::string, Arr> m;
    while (true) {
        uint32_t nl;
        if (!f.read(reinterpret_cast<char*>(&nl), 4)) break;
        std::string name(nl, ' ');
        f.read(&name[0], nl);
        uint8_t dt; uint64_t cnt;
        f.read(reinterpret_cast<char*>(&dt), 1);
        f.read(reinterpret_cast<char*>(&cnt), 8);
        const size_t es = dt == 0 ? 4 : dt == 1 ? 8 : 1;
        Arr a; a.dtype = dt; a.count = cnt; a.bytes.resize(cnt * es);
        f.read(a.bytes.data(), (std::streamsize)(cnt * es));
        m[name] = std::move(a);
    }
    return m;
}

#define CK(call) do { int rc_ = (call); if (rc_ != OPMHIP_SUCCESS) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, opmhip_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s case.bin reorder dt nsteps state_out.bin\n", argv[0]); return 2; }
    auto C = readCase(argv[1]);
    const std::string reorder = argv[2];
    const double dt = std::atof(argv[3]);
    const int nsteps = std::atoi(argv[4]);
    opmhip_config cfg;
    opmhip_default_config(&cfg);
    cfg.reorder = reorder == "level_scheduling" ? OPMHIP_REORDER_LEVEL_SCHEDULING : reorder == "graph_coloring" ? OPMHIP_REORDER_GRAPH_COLORING
                : reorder == "graph_coloring_greedy" ? OPMHIP_REORDER_GRAPH_COLORING_GREEDY : OPMHIP_REORDER_LINE_COLORING;
    opmhip_ctx* ctx = nullptr;
    if (opmhip_create(&cfg, &ctx) != OPMHIP_SUCCESS) { std::fprintf(stderr, "no device: %s\n", opmhip_last_error(nullptr)); return 77; }
    const int Nb = (int)C["rowptr"].count - 1, nnzb = (int)C["col"].count;
    CK(opmhip_set_pattern(ctx, Nb, nnzb, C["rowptr"].i32(), C["col"].i32()));
    opmhip_fluid fl{};
    const int* hdr = C["fluid_hdr"].i32();
    fl.num_pvt = hdr[0]; fl.num_sat = hdr[1];
    fl.pvtw = C["pvtw"].f64(); fl.density = C["density"].f64(); fl.pvdg_ptr = C["pvdg_ptr"].i32(); fl.pvdg = C["pvdg"].f64();
    fl.pvto_node_ptr = C["pvto_node_ptr"].i32(); fl.pvto_rs = C["pvto_rs"].f64(); fl.pvto_row_ptr = C["pvto_row_ptr"].i32(); fl.pvto = C["pvto"].f64();
    fl.swof_ptr = C["swof_ptr"].i32(); fl.swof = C["swof"].f64(); fl.sgof_ptr = C["sgof_ptr"].i32(); fl.sgof = C["sgof"].f64();
    fl.rock_pref = C["rock"].f64()[0]; fl.rock_cr = C["rock"].f64()[1];
    CK(opmhip_set_fluid(ctx, &fl));
    CK(opmhip_set_static(ctx, C["trans"].f64(), C["area"].f64(), nullptr, C["poro"].f64(), C["volume"].f64(), C["depth"].f64(), nullptr, nullptr, nullptr));
    CK(opmhip_set_state(ctx, C["pv"].f64(), C["meaning"].u8()));
    if (C.count("source")) CK(opmhip_set_source(ctx, C["source"].f64(), nullptr));
    Opm::BlackoilModelHip model(ctx, Nb, nnzb);
    Opm::SimulatorReportSingle total;
    const bool adaptive = std::getenv("OPMHIP_TEST_ADAPTIVE") != nullptr;  // report steps of length dt under the sub-step control
    double sub = dt;
    for (int s = 0; s < nsteps; ++s) {
        try {
            Opm::SimulatorReportSingle r;
            if (adaptive) {
                int chopped = 0;
                sub = model.advanceReportStep(dt, sub, r, &chopped);
                std::printf("report step %d chopped %d next dt %.6g\n", s, chopped, sub);
            } else {
                r = model.step(dt);
            }
            total += r;
            std::printf("step %d newton %u linear %u\n", s, r.total_newton_iterations, r.total_linear_iterations);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "step %d failed: %s\n", s, e.what());
            return 4;
        }
    }
    std::printf("total newton %u linear %u linearizations %u assemble %.6f setup %.6f solve %.6f update %.6f\n", total.total_newton_iterations,
                total.total_linear_iterations, total.total_linearizations, total.assemble_time, total.linear_solve_setup_time,
                total.linear_solve_time, total.update_time);
    std::vector<double> pv((size_t)Nb * 3);
    std::vector<unsigned char> mean(Nb);
    CK(opmhip_get_state(ctx, pv.data(), mean.data()));
    std::ofstream o(argv[5], std::ios::binary);
    o.write(reinterpret_cast<const char*>(pv.data()), (std::streamsize)(pv.size() * 8));
    o.write(reinterpret_cast<const char*>(mean.data()), (std::streamsize)mean.size());
    opmhip_destroy(ctx);
    return 0;
}
