"""ctypes front for libsynthframes.so — the exact-integer synthetic frame / descriptor generator
(bench and test input only; see csrc/synth_frames.c for the definition of the families)."""
import ctypes
import os
import numpy as np

NOISE, BLOCKS, FLAT, LOWTEX, MIDTEX, WARP = 0, 1, 2, 3, 4, 5
WARP_SEQ = 64              # S-warp: frames index // 64 share one base texture, index % 64 is the position on the camera path
FAMILY_NAMES = {NOISE: "S-noise", BLOCKS: "S-blocks", FLAT: "S-flat", LOWTEX: "S-lowtex", MIDTEX: "S-midtex", WARP: "S-warp"}
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsynthframes.so")
        if not os.path.exists(path):
            raise RuntimeError("libsynthframes.so missing: run `make` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        L.synth_frames.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.synth_frames.restype = None
        L.synth_descriptors.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
        L.synth_descriptors.restype = None
        _LIB = L
    return _LIB


def frames(w, h, family, first_index, n, threads=1):
    """-> uint8 array [n, h, w], frame i generated with index first_index+i.  threads > 1: frames are independent, the C call
    releases the GIL (2048 VGA frames take ~6 s on one core)."""
    out = np.empty((n, h, w), dtype=np.uint8)
    L = _lib()
    if threads <= 1 or n < 2 * threads:
        L.synth_frames(out.ctypes.data, w, h, family, first_index, n)
        return out
    from concurrent.futures import ThreadPoolExecutor
    step = (n + 4 * threads - 1) // (4 * threads)

    def part(i0):
        m = min(step, n - i0)
        L.synth_frames(out[i0:].ctypes.data, w, h, family, first_index + i0, m)

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(part, range(0, n, step)))
    return out


def frame(w, h, family, index):
    return frames(w, h, family, index, 1)[0]


def descriptors(n, seed):
    """-> uint8 array [n, 32] of PRNG 256-bit descriptors."""
    out = np.empty((n, 32), dtype=np.uint8)
    _lib().synth_descriptors(out.ctypes.data, n, seed)
    return out


def vocabulary(k, L, seed=1, ragged=False, order="bfs", stop_frac=0.02, min_leaf_level=1):
    """Synthetic vocabulary tree in the reference's node-table form (what TemplatedVocabulary::loadFromTextFile reads):
    dict(k, L, parent[int32], is_leaf[uint8], desc[n,32], weight[float64]); node 0 is the root.
    ragged: nodes get 1..k children and some stop early (leaves at levels min_leaf_level..L), as k-means leaves with few
    features do.
    order: 'bfs' (level by level) or 'kmeans' (all children of a node, then recurse: the order DBoW2's create() yields)."""
    rng = np.random.default_rng(seed)
    # level-by-level construction
    parents = [np.zeros(1, np.int64)]          # per level: parent index (global bfs id)
    leaf_flags = [np.zeros(1, bool)]
    start = [0]
    total = 1
    for lev in range(1, L + 1):
        prev_ids = np.arange(start[-1], start[-1] + len(parents[-1]))
        internal = prev_ids[~leaf_flags[-1]]
        if ragged:
            cnt = rng.integers(1, k + 1, size=len(internal))
        else:
            cnt = np.full(len(internal), k)
        par = np.repeat(internal, cnt)
        if lev == L:
            lf = np.ones(len(par), bool)
        elif ragged:
            lf = (rng.random(len(par)) < 0.15) & (lev >= min_leaf_level)
            if lf.all() and len(lf):
                lf[0] = False
        else:
            lf = np.zeros(len(par), bool)
        start.append(total)
        total += len(par)
        parents.append(par)
        leaf_flags.append(lf)
    parent = np.concatenate(parents).astype(np.int32)
    is_leaf = np.concatenate(leaf_flags)
    n = len(parent)
    desc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    weight = np.zeros(n, np.float64)
    nl = int(is_leaf.sum())
    w = rng.random(nl) * 9.0 + 0.01
    w[rng.random(nl) < stop_frac] = 0.0        # stopped words (weight 0 is skipped by transform)
    weight[is_leaf] = w
    if order == "kmeans":
        # renumber: all children of a node first, then recurse into them in order
        children = [[] for _ in range(n)]
        for i in range(1, n):
            children[parent[i]].append(i)
        new_of = np.zeros(n, np.int64)
        seq = [0]
        stack = [0]
        nxt = 1
        while stack:
            t = stack.pop()
            for c in children[t]:
                new_of[c] = nxt
                nxt += 1
                seq.append(c)
            for c in reversed(children[t]):
                stack.append(c)
        seq = np.array(seq)
        parent = new_of[parent[seq]].astype(np.int32)
        is_leaf, desc, weight = is_leaf[seq], desc[seq], weight[seq]
    parent[0] = 0
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf.astype(np.uint8), desc=np.ascontiguousarray(desc), weight=weight)


def write_vocabulary_text(path, voc, scoring=0, weighting=0):
    """the reference's text format (TemplatedVocabulary::saveToTextFile): header, then 'parent isLeaf d0..d31 weight' per node;
    no newline after the last node (the reference's eof-driven loader would read one node too many)"""
    lines = ["%d %d %d %d" % (voc["k"], voc["L"], scoring, weighting)]
    p, lf, d, w = voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]
    for i in range(1, len(p)):
        lines.append("%d %d %s %s" % (p[i], 1 if lf[i] else 0, " ".join(map(str, d[i].tolist())), repr(float(w[i]))))
    with open(path, "w") as f:
        f.write("\n".join(lines))
