"""Pins the oracle's LiveOilPvt restatement on the reference's only numeric PVT fixture:
tests/test_norne_pvt.cpp:64-294 + tests/norne_pvt.data (68 (Rs, p) points over two PVTNUM regions, including
extrapolated points with negative viscosities at :226).  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle_bind


@pytest.fixture(scope="module")
def norne(pkg, golden):
    with open(os.path.join(golden, "norne_pvt.json")) as f:
        d = json.load(f)
    pvt = []
    for r, nodes in enumerate(d["pvto"]):
        dens = d["density"][r]
        pvt.append(dict(pvtw=[1e5, 1.0, 0.0, 1e-3, 0.0], density=[dens["oil"], dens["water"], dens["gas"]],
                        pvdg=[[1e5, 1.0, 1e-5], [1e7, 0.01, 2e-5]], pvto=nodes))
    sat = [dict(swof=[[0.0, 0.0, 1.0, 0.0], [1.0, 1.0, 0.0, 0.0]], sgof=[[0.0, 0.0, 1.0, 0.0], [1.0, 1.0, 0.0, 0.0]])]
    return pkg.fluid.Fluid(pvt, sat), d


@pytest.mark.parametrize("region", [0, 1])
def test_norne_live_oil_points(orc, norne, region):
    fluid, d = norne
    e = d["expected"][region]
    mu, ib, rsat = oracle_bind.oil_pvt_probe(orc, fluid, region, e["rs"], e["p"])
    tol = d["check_close_percent"] * 1e-2  # BOOST_CHECK_CLOSE takes percent
    # the expectations are printed with 9-11 significant digits: allow one unit of the last printed digit on top
    for got, want in ((mu, e["mu_expected"]), (ib, e["b_expected"])):
        want = np.array(want)
        err = np.abs(got - want) / np.abs(want)
        assert np.all(err <= tol + 2e-9), (np.argmax(err), err.max(), got[np.argmax(err)], want[np.argmax(err)])


def test_spe1_master_table_extension(pkg, orc):
    """SPE1's PVTO has 7 Rs nodes with only the saturated sample: they must inherit an undersaturated branch, so
    that undersaturated evaluation between those nodes is finite and monotone in p (B_o shrinks with pressure)."""
    fluid, d = pkg.fluid.spe1_fluid()
    rs = np.full(5, 0.5 * (d["pvto"][2]["rs"] + d["pvto"][3]["rs"]))
    p = np.linspace(1.5e7, 4.0e7, 5)
    mu, ib, rsat = oracle_bind.oil_pvt_probe(orc, fluid, 0, rs, p)
    assert np.all(rs < rsat) and np.all(np.isfinite(ib)) and np.all(np.diff(ib) > 0) and np.all(mu > 0)
    # saturated branch reproduces the table nodes
    node = d["pvto"][4]
    mu, ib, rsat = oracle_bind.oil_pvt_probe(orc, fluid, 0, [node["rs"] * 2], [node["p"][0]])
    assert abs(rsat[0] - node["rs"]) < 1e-12 * node["rs"]
    assert abs(ib[0] - 1.0 / node["bo"][0]) < 1e-14 and abs(mu[0] - node["mu"][0]) < 1e-15
