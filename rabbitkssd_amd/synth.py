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


def clade_sketches(n_genomes, m, hash_bits, kmer_size=20, strains_per_clade=10, seed=20261003):
    """Sketch-level generator: per clade draw `m` distinct uniform values in
    [0, 2^hash_bits); strain s keeps each with probability (1-0.002 s)^kmer_size and
    replaces the rest with fresh uniform values; per-genome sets are deduplicated and
    returned SORTED.  Returns (names, hashes uint32[H], off uint64[N+1])."""
    rng = np.random.default_rng(seed)
    space = 1 << hash_bits
    names, parts = [], []
    anc = None
    for g in range(n_genomes):
        c, s = divmod(g, strains_per_clade)
        if s == 0:
            anc = np.unique(rng.integers(0, space, size=m + m // 8, dtype=np.uint64))
            rng.shuffle(anc)
            anc = anc[:m]
        keep = rng.random(len(anc)) < (1.0 - 0.002 * s) ** kmer_size
        fresh = rng.integers(0, space, size=len(anc), dtype=np.uint64)
        h = np.unique(np.where(keep, anc, fresh)).astype(np.uint32)
        parts.append(h)
        names.append("syn/c%04d_s%d.fna" % (c, s))
    off = np.zeros(n_genomes + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    hashes = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    return names, hashes, off
