"""Sequence helpers: residue codes (letter - 'A', /root/reference/sift4g/src/hash.cpp:16-19) and FASTA."""
import numpy as np


def encode(seq):
    """str -> uint8 codes 0..25; non-letters are dropped, case folded."""
    b = np.frombuffer(seq.upper().encode("ascii", "ignore"), dtype=np.uint8)
    b = b[(b >= 65) & (b <= 90)]
    return (b - 65).astype(np.uint8)


def decode(codes):
    return (np.asarray(codes, dtype=np.uint8) + 65).tobytes().decode("ascii")


def read_fasta(path):
    """-> list of (name, codes). Name = header text after '>' (whole line), as sift4g uses it for file stems."""
    out, name, parts = [], None, []
    with open(path, "r") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, encode("".join(parts))))
                name, parts = line[1:], []
            elif name is not None:
                parts.append(line)
    if name is not None:
        out.append((name, encode("".join(parts))))
    return out


def flatten(seqs):
    """list of uint8 arrays -> (codes uint8[total], offsets uint64[n+1])"""
    n = len(seqs)
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum(np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=n))
    total = int(off[-1])
    codes = np.empty(max(total, 1), dtype=np.uint8)
    if total:
        codes[:total] = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs])
    return codes, off


def write_fasta(path, names, codes, offsets, width=60):
    """writes sequences given as concatenated codes + offsets"""
    with open(path, "w") as fh:
        letters = (np.asarray(codes, dtype=np.uint8) + 65).tobytes().decode("ascii")
        for i, name in enumerate(names):
            s = letters[int(offsets[i]):int(offsets[i + 1])]
            fh.write(f">{name}\n")
            for k in range(0, len(s), width):
                fh.write(s[k:k + width] + "\n")
