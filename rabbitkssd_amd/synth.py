"""Deterministic synthetic inputs: clade-model FASTA genomes and
sketch-level clade sets.  Used by tests, bench.py and the golden-fixture script; there is
no network, so every workload is generated from fixed seeds."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def clade_genome(clade, strain, length, rate_per_strain=0.002):
    """ASCII bases (np.uint8) of strain `strain` of clade `clade`: the clade ancestor is
    i.i.d. uniform ACGT from PRNG(1000+clade); the strain substitutes each base with
    probability rate_per_strain*strain from PRNG(5000+10*clade+strain)."""
    anc = np.random.default_rng(1000 + clade).integers(0, 4, size=length, dtype=np.uint8)
    if strain:
        rng = np.random.default_rng(5000 + 10 * clade + strain)
        mut = rng.random(length) < rate_per_strain * strain
        shift = rng.integers(1, 4, size=length, dtype=np.uint8)
        anc = np.where(mut, (anc + shift) & 3, anc).astype(np.uint8)
    return _ACGT[anc]


def fasta_text(name, bases, width=80):
    """One-record FASTA file text (bytes) with `width`-column lines."""
    n = len(bases)
    full = n // width
    body = bases[: full * width].reshape(full, width)
    lines = np.concatenate([body, np.full((full, 1), 10, dtype=np.uint8)], axis=1).tobytes()
    tail = bases[full * width:].tobytes()
    if tail:
        tail += b"\n"
    return b">" + name.encode() + b"\n" + lines + tail


def clade_genome_set(n_genomes, length, strains_per_clade=10):
    """[(name, bases)] for genomes c{clade}_s{strain}, clades of `strains_per_clade`."""
    out = []
    for g in range(n_genomes):
        c, s = divmod(g, strains_per_clade)
        out.append(("c%d_s%d" % (c, s), clade_genome(c, s, length)))
    return out


def strain_rates(s):
    """Per-level substitution rates of strain `s` of a clade (species).  Up to 10 strains: the SURVEY 8d model, strain s
    differs from the ancestor by 0.002 s.  Wider clades are a tree: s = 100 a + 10 b + c is strain c of sub-lineage b of
    lineage a; a lineage differs from the ancestor by 0.026, a sub-lineage from its lineage by 0.006, a strain from its
    sub-lineage by 0.001 c.  Mash distances: inside a sub-lineage <= 0.018, inside a lineage 0.012-0.030 (all reportable
    at -D 0.05), across lineages >= 0.064 (never reportable, yet such pairs still share ~28 % of their hashes)."""
    a, b, c = s // 100, (s // 10) % 10, s % 10
    return a, b, c


def clade_sketches(n_genomes, m, hash_bits, kmer_size=20, strains_per_clade=10, seed=20261003, tiny=0, tiny_size=40, wide=False):
    """Sketch-level generator: per clade draw `m` distinct uniform values in
    [0, 2^hash_bits); strain s keeps each with probability (1-0.002 s)^kmer_size and
    replaces the rest with fresh uniform values; per-genome sets are deduplicated and
    returned SORTED.  Returns (names, hashes uint32[H], off uint64[N+1]).
    strains_per_clade > 10: a species tree of lineages / sub-lineages / strains (strain_rates).
    tiny: that many extra sketches of `tiny_size` hashes (plasmids, small contigs) are spread over the collection; each is a
    random subset of one clade ancestor (so it has relatives and reportable pairs under containment).
    wide: hashes as uint64 (the 64-bit layout of half_k - drlevel > 8, e.g. K12 L3: 36 bits)."""
    rng = np.random.default_rng(seed)
    space = 1 << hash_bits
    names, parts = [], []
    anc = None
    tree = strains_per_clade > 10

    def mutate(src, rate):
        keep = rng.random(len(src)) < (1.0 - rate) ** kmer_size
        fresh = rng.integers(0, space, size=len(src), dtype=np.uint64)
        return np.where(keep, src, fresh)

    lineage = sub = None
    for g in range(n_genomes):
        c, s = divmod(g, strains_per_clade)
        if s == 0:
            anc = np.unique(rng.integers(0, space, size=m + m // 8, dtype=np.uint64))
            rng.shuffle(anc)
            anc = anc[:m]
        if not tree:
            h = mutate(anc, 0.002 * s)
        else:
            a, b, cc = strain_rates(s)
            if s % 100 == 0:
                lineage = mutate(anc, 0.026)
            if s % 10 == 0:
                sub = mutate(lineage, 0.006)
            h = mutate(sub, 0.001 * cc)
        parts.append(np.unique(h).astype(np.uint64 if wide else np.uint32))
        names.append("syn/c%04d_s%d.fna" % (c, s))
        if tiny and (g + 1) % max(1, n_genomes // tiny) == 0 and len(names) - (g + 1) < tiny:
            t = np.unique(anc[rng.permutation(len(anc))[:tiny_size]]).astype(np.uint64 if wide else np.uint32)
            parts.append(t)
            names.append("syn/tiny%04d.fna" % (len(names) - (g + 1)))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    hashes = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64 if wide else np.uint32)
    return names, hashes, off


def permute_genomes(names, hashes, off, order):
    """The same sketches listed in another order: genome i of the result is genome order[i] of the input."""
    off = np.asarray(off, dtype=np.uint64)
    order = np.asarray(order, dtype=np.int64)
    sizes = np.diff(off).astype(np.int64)[order]
    new_off = np.zeros(len(order) + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(sizes)
    # gather index of every element: start of its source genome + position inside it
    starts = np.repeat(off[:-1].astype(np.int64)[order] - new_off[:-1].astype(np.int64), sizes)
    idx = starts + np.arange(int(new_off[-1]), dtype=np.int64)
    return [names[i] for i in order], np.ascontiguousarray(hashes[idx]), new_off


def genome_order(n, mode, seed=7, window=256):
    """Orders in which a collection can arrive: "sorted" (as generated: clade members adjacent), "shuffled" (a uniform
    random permutation of the genome ids), "jitter" (OpenMP completion order, src/sketch.cpp:558-568: a random
    permutation inside every window of `window` consecutive genomes)."""
    rng = np.random.default_rng(seed)
    if mode == "sorted":
        return np.arange(n, dtype=np.int64)
    if mode == "shuffled":
        return rng.permutation(n).astype(np.int64)
    if mode == "jitter":
        order = np.arange(n, dtype=np.int64)
        for a in range(0, n, window):
            order[a:a + window] = a + rng.permutation(min(window, n - a))
        return order
    raise ValueError(mode)


def shuf_table(half_k, half_subk, drlevel):
    """The .shuf dimension table (int32[16^half_subk]) for these parameters, generated by the product's own
    `rabbit_kssd shuffle` (host code, src/shuffle.cpp:25-104 semantics) into a temporary file."""
    import os
    import subprocess
    import tempfile
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rabbit_kssd")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.shuf")
        subprocess.run([tool, "shuffle", "-k", str(half_k), "-s", str(half_subk), "-l", str(drlevel), "-o", path],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = np.fromfile(path, dtype=np.int32)
    return raw[4:].copy()  # 16-byte header {id, k, subk, drlevel}, src/shuffle.cpp:8-23


def write_sketch_file(path, half_k, half_subk, drlevel, names, hashes, off):
    """A RabbitKSSD .sketch file (saveSketches, src/sketch.cpp:1024-1068) from a CSR of
    32-bit hashes: sketchInfo_t {id, half_k, half_subk, drlevel, genomeNumber}, int32 nameLen[N], int32 hashCount[N],
    then per genome the name bytes followed by its hashes."""
    n = len(names)
    off = np.asarray(off, dtype=np.uint64)
    hashes = np.ascontiguousarray(hashes, dtype="<u4")
    enc = [s.encode() for s in names]
    info = np.array([(half_k << 8) + (half_subk << 4) + drlevel, half_k, half_subk, drlevel, n], dtype="<i4")
    with open(path, "wb") as f:
        f.write(info.tobytes())
        f.write(np.array([len(e) for e in enc], dtype="<i4").tobytes())
        f.write(np.diff(off).astype("<i4").tobytes())
        raw = hashes.view(np.uint8)
        for g in range(n):
            f.write(enc[g])
            f.write(raw[4 * int(off[g]):4 * int(off[g + 1])].tobytes())


# ---- large collections, generated on the device (round 5: the scale leg of bench.py) ----------------------------------
def _mix64(torch, x):
    """splitmix64 finaliser on int64 tensors (a counter-based generator: value = f(stream, id, position), nothing is stored)"""
    x = (x ^ (x >> 30)) * -4658895280553007687   # 0xBF58476D1CE4E5B9
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685   # 0x94D049BB133111EB
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def scale_species_plan(n_genomes, seed=20261005, max_species=10000, zipf_a=2.0, min_size=200, max_size=3000):
    """species of a large collection: sizes Zipf-distributed (1 .. max_species strains), every species with its own sketch
    size, log-uniform in [min_size, max_size] (a sketch size follows the genome length: strains of a species agree).
    Returns (species id of every genome int64[N], sketch size of every species int64[S])."""
    rng = np.random.default_rng(seed)
    sizes = []
    left = n_genomes
    while left > 0:
        s = int(min(max_species, rng.zipf(zipf_a), left))
        sizes.append(s)
        left -= s
    sizes = np.array(sizes, dtype=np.int64)
    m = np.exp(rng.uniform(np.log(min_size), np.log(max_size), size=len(sizes))).astype(np.int64)
    return np.repeat(np.arange(len(sizes), dtype=np.int64), sizes), m


def scale_collection_torch(n_genomes, hash_bits=28, kmer_size=20, seed=20261005, device="cuda", max_species=10000, chunk_elems=1 << 26):
    """A collection of `n_genomes` sketches generated ON THE DEVICE (torch): species by scale_species_plan; inside a species the
    tree of strain_rates (lineages of 100 strains 0.026 from the ancestor, sub-lineages of 10 strains 0.006 from their lineage,
    strain c 0.001 c from its sub-lineage); a hash of the level above survives with probability (1 - rate)^k, else it is replaced
    by a fresh uniform value -- every value a function of (level, id, position) through a counter-based generator.  Per genome
    the hashes are sorted and distinct.  Returns (hashes int32 tensor viewed as u32, off int64 tensor[N + 1], species int64[N])."""
    import torch
    species_np, m_np = scale_species_plan(n_genomes, seed, max_species)
    species = torch.from_numpy(species_np).to(device)
    m = torch.from_numpy(m_np).to(device)
    first = torch.zeros(len(m_np), dtype=torch.int64, device=device)   # first genome of every species
    first[1:] = torch.cumsum(torch.bincount(species, minlength=len(m_np)), 0)[:-1]
    size = m[species]
    rank = torch.arange(n_genomes, device=device) - first[species]      # strain number inside its species
    space_mask = (1 << hash_bits) - 1
    p_lin, p_sub = (1.0 - 0.026) ** kmer_size, (1.0 - 0.006) ** kmer_size

    def rnd(stream, ident, pos):
        return _mix64(torch, _mix64(torch, ident * 1000003 + stream * 7919 + seed) ^ (pos * -7046029254386353131))

    def keep(r, p):   # p: float or tensor
        return ((r >> 11) & 0xFFFFFF).to(torch.float32) < (p * 16777216.0)

    out_h, sizes = [], torch.zeros(n_genomes, dtype=torch.int64, device=device)
    start = 0
    csum = torch.cumsum(size, 0)
    while start < n_genomes:   # chunks of whole genomes
        base = int(csum[start - 1].item()) if start else 0
        stop = int(torch.searchsorted(csum, torch.tensor(base + chunk_elems, device=device)).item())
        stop = max(start + 1, min(n_genomes, stop))
        g = torch.repeat_interleave(torch.arange(start, stop, device=device), size[start:stop])
        pos = torch.arange(len(g), device=device) - torch.repeat_interleave(csum[start:stop] - size[start:stop] - base, size[start:stop])
        sp, rk = species[g], rank[g]
        lin, sub = sp * 1024 + rk // 100, sp * 16384 + rk // 10
        v = rnd(1, sp, pos)                                             # the species' ancestor
        v = torch.where(keep(rnd(2, lin, pos), p_lin), v, rnd(3, lin, pos))
        v = torch.where(keep(rnd(4, sub, pos), p_sub), v, rnd(5, sub, pos))
        p_str = torch.pow(1.0 - 0.001 * (rk % 10).to(torch.float32), kmer_size)
        v = torch.where(keep(rnd(6, g, pos), p_str), v, rnd(7, g, pos))
        key = (g << hash_bits) | (v & space_mask)
        key = torch.unique(key)                                         # sorted by (genome, hash), repeats dropped
        gg = key >> hash_bits
        sizes[start:stop] = torch.bincount(gg - start, minlength=stop - start)
        out_h.append((key & space_mask).to(torch.int32))
        del g, pos, sp, rk, lin, sub, v, key, gg
        start = stop
    hashes = torch.cat(out_h)
    off = torch.zeros(n_genomes + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(sizes, 0)
    return hashes, off, species


def canonical_skew(hashes, off, hash_bits, levels=1):
    """The hash of a real sketch is a piece of a CANONICAL k-mer -- the smaller of a k-mer and its reverse complement --, and its top
    bits are that k-mer's leading bases: A leads seven times as often as T (7 : 5 : 3 : 1 for A, C, G, T in random sequence; the
    sketcher's own output for 1,000 random 5 Mb genomes: tools/hash_dist_probe.py).  Maps a uniform collection monotonically onto
    quarters of the hash space filled 7 : 5 : 3 : 1 (`levels` = 2: once more inside every quarter) and drops the repeats the map
    creates inside a sketch.  Returns (hashes uint32, off uint64)."""
    n = len(off) - 1
    x = hashes.astype(np.float64) / float(1 << hash_bits)
    cum = np.array([0.0, 7.0, 12.0, 15.0, 16.0]) / 16.0

    def quarters(u):
        q = np.minimum(3, np.searchsorted(cum, u, side="right") - 1)
        return (q + (u - cum[q]) / (cum[q + 1] - cum[q])) / 4.0
    y = quarters(x)
    if levels > 1:
        y = (np.floor(y * 4.0) + quarters((y * 4.0) % 1.0)) / 4.0
    hs = np.minimum((1 << hash_bits) - 1, np.floor(y * float(1 << hash_bits))).astype(np.uint32)
    gid = np.repeat(np.arange(n, dtype=np.int64), np.diff(off).astype(np.int64))
    key = (gid << hash_bits) | hs.astype(np.int64)
    keep = np.concatenate(([True], key[1:] != key[:-1])) if len(key) else np.zeros(0, dtype=bool)
    hs, gid = hs[keep], gid[keep]
    off2 = np.zeros(n + 1, dtype=np.uint64)
    off2[1:] = np.cumsum(np.bincount(gid, minlength=n))
    return hs, off2

