"""MatrixMarket reader for dune-istl "ISTL_STRUCT blocked" files.

The reference reads its solver fixtures with Dune's readMatrixMarket
(tests/test_cusparseSolver.cpp:59-73); this is the same file format turned into the block-CSR
triplet the accelerator boundary uses (bda/BdaBridge.cpp:167-189: rows = Nb+1 block row pointers,
cols = ascending block columns, vals = row-major bs*bs blocks).
"""
import numpy as np


def _header(lines):
    bs = (1, 1)
    i = 0
    kind = lines[0].split()[2]  # coordinate | array
    while lines[i].startswith("%"):
        if "ISTL_STRUCT" in lines[i]:
            t = lines[i].split()
            bs = (int(t[-2]), int(t[-1]))
        i += 1
    return kind, bs, i


def read_block_matrix(path):
    """-> (Nb, rowptr[int32], colidx[int32], vals[float64 nnzb*bs*bs], bs)"""
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    kind, (br, bc), i = _header(lines)
    assert kind == "coordinate" and br == bc
    bs = br
    n, m, nnz = (int(t) for t in lines[i].split())
    ent = np.array([l.split() for l in lines[i + 1:i + 1 + nnz]], dtype=np.float64)
    r = ent[:, 0].astype(np.int64) - 1
    c = ent[:, 1].astype(np.int64) - 1
    v = ent[:, 2]
    Nb = n // bs
    blocks = {}
    for rr, cc, vv in zip(r, c, v):
        key = (int(rr // bs), int(cc // bs))
        blk = blocks.setdefault(key, np.zeros((bs, bs)))
        blk[rr % bs, cc % bs] = vv
    keys = sorted(blocks)
    rowptr = np.zeros(Nb + 1, dtype=np.int32)
    for (bi, _) in keys:
        rowptr[bi + 1] += 1
    rowptr = np.cumsum(rowptr).astype(np.int32)
    colidx = np.array([k[1] for k in keys], dtype=np.int32)
    vals = np.concatenate([blocks[k].reshape(-1) for k in keys]).astype(np.float64)
    return Nb, rowptr, colidx, vals, bs


def read_block_vector(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    kind, _, i = _header(lines)
    assert kind == "array"
    n, _ = (int(t) for t in lines[i].split())
    return np.array([float(l) for l in lines[i + 1:i + 1 + n]], dtype=np.float64)
