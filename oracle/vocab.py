"""DBoW3 binary vocabulary files on the oracle side.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

An independent statement (pure Python) of the container the library's C++ reader parses: DBoW3::Vocabulary::toStream's
layout and the QuickLZ 1.5 level-1 packets it wraps the payload in, plus a writer of synthetic vocabularies (stored
packets only) for tests that must run where the reference's Vocabulary/ files are absent.  DBoW3 is a third-party
dependency the reference does not vendor (CMakeLists.txt:16-17): parity against DBoW3 itself is unpinned."""
import struct

import numpy as np

SIGNATURE = 88877711233


def _rd(buf, p, n):
    b = bytes(buf[p:p + n])
    return int.from_bytes(b + b"\0" * (n - len(b)), "little")


def qlz_decompress(src):
    """One QuickLZ 1.5 packet (level 1, no streaming) -> (payload bytes, packet length)."""
    fl = src[0]
    if fl & 2:
        cs, ds = struct.unpack_from("<II", src, 1)
        hdr = 9
    else:
        cs, ds, hdr = src[1], src[2], 3
    if not fl & 1:
        return bytes(src[hdr:hdr + ds]), cs
    if (fl >> 2) & 3 != 1:
        raise ValueError("only QuickLZ level 1 is read")
    dst = bytearray(ds)
    hashtab = [0] * 4096
    s, d, last = hdr, 0, ds - 1
    last_matchstart = last - 6 - 4
    last_hashed = -1
    cword = 1
    bitlut = [4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0]

    def hash_upto(upto):
        nonlocal last_hashed
        while last_hashed < upto:
            last_hashed += 1
            f = _rd(dst, last_hashed, 3)
            hashtab[((f >> 12) ^ f) & 4095] = last_hashed

    while True:
        if cword == 1:
            cword = _rd(src, s, 4)
            s += 4
        fetch = _rd(src, s, 4)
        if cword & 1:
            cword >>= 1
            off = hashtab[(fetch >> 4) & 0xfff]
            if fetch & 0xf:
                ml = (fetch & 0xf) + 2
                s += 2
            else:
                ml = src[s + 2]
                s += 3
            for i in range(ml):
                dst[d + i] = dst[off + i]
            d += ml
            hash_upto(d - ml)
            last_hashed = d - 1
        elif d < last_matchstart:
            n = bitlut[cword & 0xf]
            dst[d:d + n] = src[s:s + n]
            cword >>= n
            d += n
            s += n
            hash_upto(d - 3)
        else:
            while d <= last:
                if cword == 1:
                    s += 4
                    cword = 1 << 31
                dst[d] = src[s]
                d += 1
                s += 1
                cword >>= 1
            return bytes(dst), cs


def load(path):
    raw = open(path, "rb").read()
    sig, = struct.unpack_from("<Q", raw, 0)
    if sig != SIGNATURE:
        raise ValueError("not a DBoW3 binary vocabulary")
    compressed = raw[8] != 0
    nnodes, = struct.unpack_from("<I", raw, 9)
    if compressed:
        nch, = struct.unpack_from("<I", raw, 13)
        o, parts, n_packed = 17, [], 0
        for _ in range(nch):
            n_packed += raw[o] & 1
            part, used = qlz_decompress(raw[o:])
            parts.append(part)
            o += used
        data = b"".join(parts)
    else:
        data, n_packed = raw[13:], 0
    k, L, scoring, weighting = struct.unpack_from("<iiii", data, 0)
    p = 16
    parent = np.full(nnodes, -1, np.int32)
    weight = np.zeros(nnodes, np.float64)
    word_id = np.full(nnodes, -1, np.int32)
    children = [[] for _ in range(nnodes)]
    desc = None
    for _ in range(1, nnodes):
        nid, par = struct.unpack_from("<II", data, p)
        w, = struct.unpack_from("<d", data, p + 8)
        cols, rows, typ = struct.unpack_from("<iii", data, p + 16)
        assert (rows, typ) == (1, 5)
        if desc is None:
            desc = np.zeros((nnodes, cols), np.float32)
        desc[nid] = np.frombuffer(data, np.float32, cols, p + 28)
        p += 28 + 4 * cols
        parent[nid], weight[nid] = par, w
        children[par].append(nid)
    nw, = struct.unpack_from("<I", data, p)
    p += 4
    word_node = np.full(nw, -1, np.int32)
    for _ in range(nw):
        wid, nid = struct.unpack_from("<II", data, p)
        p += 8
        word_id[nid] = wid
        word_node[wid] = nid
    assert p == len(data)
    child_off = np.zeros(nnodes + 1, np.int32)
    child_off[1:] = np.cumsum([len(c) for c in children])
    child_idx = np.array([c for cs in children for c in cs], np.int32)
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, n_nodes=nnodes, n_words=nw, dim=desc.shape[1], parent=parent,
                weight=weight, word_id=word_id, child_off=child_off, child_idx=child_idx, word_node=word_node, desc=desc,
                packed_chunks=n_packed)


def write_synthetic(path, k=9, L=3, dim=256, seed=0, scoring=1, weighting=0, shuffle=True):
    """A k-ary tree of depth L with random unit centroids (children scattered around their parent), weights on the
    leaves, written as DBoW3 writes it: QuickLZ packets of 10000 payload bytes, all stored (a packet that does not
    shrink is stored verbatim, which is what happens to float data)."""
    rng = np.random.default_rng(seed)
    nodes = [(0, -1, np.zeros(dim, np.float32))]
    level = [0]
    for depth in range(L):
        nxt = []
        for par in level:
            for _ in range(k):
                base = nodes[par][2]
                v = base + rng.normal(size=dim).astype(np.float32) * np.float32(0.9 ** depth / 16 if depth else 1 / 16)
                v = (v / np.linalg.norm(v)).astype(np.float32)
                nodes.append((len(nodes), par, v))
                nxt.append(len(nodes) - 1)
        level = nxt
    leaves = level
    order = list(range(1, len(nodes)))
    if shuffle:   # DBoW3 lists nodes in creation order, which is not id order
        order = [int(i) for i in rng.permutation(order)]
        # children of one parent must keep their relative order across the shuffle for the tree to be the same tree
    weights = {n: float(rng.uniform(0.2, 3.0)) for n in leaves}
    for n in rng.choice(leaves, size=max(1, len(leaves) // 50), replace=False):
        weights[int(n)] = 0.0   # words that occurred in every training image
    pay = struct.pack("<iiii", k, L, scoring, weighting)
    for n in order:
        nid, par, v = nodes[n]
        pay += struct.pack("<IId", nid, par, weights.get(nid, 0.0)) + struct.pack("<iii", dim, 1, 5) + v.tobytes()
    pay += struct.pack("<I", len(leaves))
    for wid, nid in enumerate(leaves):
        pay += struct.pack("<II", wid, nid)
    out = struct.pack("<QBI", SIGNATURE, 1, len(nodes))
    chunks = [pay[i:i + 10000] for i in range(0, len(pay), 10000)]
    out += struct.pack("<I", len(chunks))
    for c in chunks:
        out += struct.pack("<BII", 0x46, len(c) + 9, len(c)) + c
    open(path, "wb").write(out)


def bow_vectors(voc, word_id, weight, node_id):
    """BowVector and FeatureVector as Vocabulary::transform(features, v, fv, levelsup) fills them."""
    bow, fv = {}, {}
    tf = voc["weighting"] in (0, 1)
    for i, (w, wt, nd) in enumerate(zip(word_id, weight, node_id)):
        if w < 0 or not wt > 0:
            continue
        if w not in bow:
            bow[int(w)] = float(wt)
        elif tf:
            bow[int(w)] += float(wt)
        fv.setdefault(int(nd), []).append(i)
    norm = 2 if voc["scoring"] == 1 else (0 if voc["scoring"] == 5 else 1)
    if tf and norm == 0 and bow:
        nd = float(len(bow))
        bow = {k: v / nd for k, v in bow.items()}
    if norm:
        s = 0.0
        for k in sorted(bow):
            s += abs(bow[k]) if norm == 1 else bow[k] * bow[k]
        if norm == 2:
            s = s ** 0.5
        if s > 0:
            bow = {k: v / s for k, v in bow.items()}
    return sorted(bow.items()), sorted(fv.items())
