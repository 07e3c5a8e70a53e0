#!/usr/bin/env python3
"""Counts, in the gfx950 ISA of fft_quad_kernel, the VMEM instructions between the LAST LDS-DMA piece of the transform loop and
the `s_waitcnt vmcnt(n)` that waits for the pieces (cyberether_amd/csrc/kernels/fft_quad.hh: JST_WAIT_VMCNT(kAfter * STORES)):
LDS-DMA and stores retire in order, so the pieces have landed when at most as many instructions as were issued behind the
last piece are outstanding -- n must equal that count, and nothing else (a scratch reload, a compiler-inserted load) may sit
in between.  Usage: check_quad_isa.py <file.s>  (hipcc -S --cuda-device-only of a unit that instantiates the kernel)."""
import re
import sys


def check(text: str) -> dict:
    start = text.index("\n_ZN3jst3dev15fft_quad_kernel")
    body = text[start:]
    body = body[:body.index(".Lfunc_end")]
    lines = body.split("\n")
    pieces = [i for i, l in enumerate(lines) if re.search(r"buffer_load_dwordx4 .* lds\s*$", l)]
    assert len(pieces) == 18, f"expected 9 prologue + 9 loop pieces, found {len(pieces)}"
    last = pieces[-1]
    waits = [(i, int(m.group(1))) for i, l in enumerate(lines) if (m := re.search(r"s_waitcnt vmcnt\((\d+)\)", l)) and i > last]
    assert waits, "no vmcnt wait behind the last piece"
    wait_line, n = waits[0]
    between = lines[last + 1:wait_line]
    stores = sum(1 for l in between if re.search(r"^\s*buffer_store_", l))
    other = [l.strip() for l in between if re.search(r"^\s*(scratch_|global_|flat_|buffer_load|buffer_atomic)", l)]
    scratch = [l.strip() for l in lines if "scratch_" in l]
    return {"vmcnt": n, "stores_behind_last_piece": stores, "other_vmem_between": other, "scratch_ops": len(scratch)}


if __name__ == "__main__":
    r = check(open(sys.argv[1]).read())
    print(r)
    ok = r["vmcnt"] == r["stores_behind_last_piece"] and not r["other_vmem_between"]
    sys.exit(0 if ok else 1)
