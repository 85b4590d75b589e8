"""Deck-level fluid / saturation-function tables in the flat SI layout the C-ABI takes (opmhip_fluid in
include/opmhip.h).  The tables are what a deck provides through PVTW, PVDG, PVTO, DENSITY, ROCK, SWOF, SGOF
(python/test_data/SPE1CASE1/SPE1CASE1.DATA:109-250); interpolation structures are built inside the library.
"""
import ctypes as C
import json
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class FluidDesc(C.Structure):
    """Mirror of `struct opmhip_fluid` (and of the oracle's identical `orc_fluid_desc`)."""
    _fields_ = [("num_pvt", C.c_int), ("num_sat", C.c_int),
                ("pvtw", C.c_void_p), ("density", C.c_void_p),
                ("pvdg_ptr", C.c_void_p), ("pvdg", C.c_void_p),
                ("pvto_node_ptr", C.c_void_p), ("pvto_rs", C.c_void_p), ("pvto_row_ptr", C.c_void_p), ("pvto", C.c_void_p),
                ("swof_ptr", C.c_void_p), ("swof", C.c_void_p), ("sgof_ptr", C.c_void_p), ("sgof", C.c_void_p),
                ("rock_pref", C.c_double), ("rock_cr", C.c_double),
                ("pvtg_node_ptr", C.c_void_p), ("pvtg_pg", C.c_void_p), ("pvtg_row_ptr", C.c_void_p), ("pvtg", C.c_void_p),
                ("num_rock", C.c_int), ("rocktab_ptr", C.c_void_p), ("rocktab", C.c_void_p),
                ("pc_scaling", C.c_int)]


class Fluid:
    """pvt: list of dict(pvtw[5], density[3] (oil, water, gas), pvdg rows (p,Bg,mu), pvto nodes dict(rs,p[],bo[],mu[]),
    optional pvtg nodes dict(pg, rv[], bg[], mu[]) - wet gas, rows as in the deck: saturated first, Rv descending - in which
    case pvdg may be omitted); sat: list of dict(swof rows (Sw,krw,krow,pcow), sgof rows (Sg,krg,krog,pcog));
    rocktab: optional list (one per rock region) of rows (p, pore-volume multiplier, transmissibility multiplier); all SI.
    pc_scaling: the deck scales the oil-water capillary pressure per cell (PCW or SWATINIT): set_pcw may then follow."""

    def __init__(self, pvt, sat, rock_pref=1e5, rock_cr=0.0, rocktab=None, pc_scaling=False):
        self.pvt, self.sat, self.rock_pref, self.rock_cr = pvt, sat, float(rock_pref), float(rock_cr)
        self.rocktab = rocktab or []
        self.pc_scaling = bool(pc_scaling)
        self.wet_gas = bool(pvt and pvt[0].get("pvtg"))
        f64 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))
        a = {}
        a["pvtw"] = f64([r["pvtw"] for r in pvt])
        a["density"] = f64([r["density"] for r in pvt])
        a["pvdg_ptr"] = i32(np.concatenate([[0], np.cumsum([len(r.get("pvdg", [])) for r in pvt])]))
        a["pvdg"] = f64(np.concatenate([np.asarray(r.get("pvdg", []), float).reshape(-1, 3) for r in pvt] + [np.zeros((0, 3))]))
        if self.wet_gas:
            gn = [n for r in pvt for n in r["pvtg"]]
            a["pvtg_node_ptr"] = i32(np.concatenate([[0], np.cumsum([len(r["pvtg"]) for r in pvt])]))
            a["pvtg_pg"] = f64([n["pg"] for n in gn])
            a["pvtg_row_ptr"] = i32(np.concatenate([[0], np.cumsum([len(n["rv"]) for n in gn])]))
            a["pvtg"] = f64(np.concatenate([np.stack([n["rv"], n["bg"], n["mu"]], axis=1) for n in gn]))
        if self.rocktab:
            a["rocktab_ptr"] = i32(np.concatenate([[0], np.cumsum([len(t) for t in self.rocktab])]))
            a["rocktab"] = f64(np.concatenate([np.asarray(t, float).reshape(-1, 3) for t in self.rocktab]))
        a["pvto_node_ptr"] = i32(np.concatenate([[0], np.cumsum([len(r["pvto"]) for r in pvt])]))
        nodes = [n for r in pvt for n in r["pvto"]]
        a["pvto_rs"] = f64([n["rs"] for n in nodes])
        a["pvto_row_ptr"] = i32(np.concatenate([[0], np.cumsum([len(n["p"]) for n in nodes])]))
        a["pvto"] = f64(np.concatenate([np.stack([n["p"], n["bo"], n["mu"]], axis=1) for n in nodes]))
        a["swof_ptr"] = i32(np.concatenate([[0], np.cumsum([len(s["swof"]) for s in sat])]))
        a["swof"] = f64(np.concatenate([np.asarray(s["swof"], float).reshape(-1, 4) for s in sat]))
        a["sgof_ptr"] = i32(np.concatenate([[0], np.cumsum([len(s["sgof"]) for s in sat])]))
        a["sgof"] = f64(np.concatenate([np.asarray(s["sgof"], float).reshape(-1, 4) for s in sat]))
        self.arrays = a

    def desc(self):
        d = FluidDesc()
        d.num_pvt, d.num_sat = len(self.pvt), len(self.sat)
        for k, v in self.arrays.items():
            setattr(d, k, v.ctypes.data_as(C.c_void_p))
        d.rock_pref, d.rock_cr = self.rock_pref, self.rock_cr
        d.num_rock = len(self.rocktab)
        d.pc_scaling = int(self.pc_scaling)
        return d


def spe1_fluid():
    """The SPE1CASE1 fluid (one PVT region, one saturation region), SI."""
    with open(os.path.join(DATA, "spe1_fluid.json")) as f:
        d = json.load(f)
    w = d["pvtw"]
    pvt = [dict(pvtw=[w["p_ref"], w["bw_ref"], w["cw"], w["mu_ref"], w["cv"]],
                density=[d["density"]["oil"], d["density"]["water"], d["density"]["gas"]], pvdg=d["pvdg"], pvto=d["pvto"])]
    sat = [dict(swof=d["swof"], sgof=d["sgof"])]
    return Fluid(pvt, sat, rock_pref=d["rock"]["p_ref"], rock_cr=d["rock"]["cr"]), d
