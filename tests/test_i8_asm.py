"""The integer pre-smoothing pass (glava_amd/csrc/glv_misc.hip glv_bars_rows_i8_kernel) requests its weight fragments with inline-asm loads and
awaits them with hand-placed `s_waitcnt vmcnt(6)` -- the compiler does not know those registers are in flight.  That is only sound if nothing
but the MFMAs (behind the wait) ever reads them: a copy the register allocator slipped in between request and wait would read stale data, and
only sometimes.  This test compiles the translation unit to assembly with the product's flags (hipcc cross-compiles, no GPU) and checks
every instantiation of the kernel: the asynchronously loaded registers are read by v_mfma instructions only and written by the asm loads
only; no scratch (spills), no FLAT memory instructions (a pointer that loses its address space makes every wait a wait for everything), at
most 256 registers (two waves per SIMD)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_fragments_of_the_integer_pass_are_touched_by_the_matrix_cores_only(tmp_path):
    from glava_amd import build as B
    src = os.path.join(ROOT, "glava_amd", "csrc", "glv_misc.hip")
    out = str(tmp_path / "glv_misc.s")
    flags = [f for f in B.HIPFLAGS if f != "-fPIC"]
    subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-S", src, "-o", out], check=True, capture_output=True)
    text = open(out).read()
    kernels = re.findall(r"^(_ZN3glv23glv_bars_rows_i8_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert len(kernels) >= 20, len(kernels)                       # 5 rings x {texel, float rows} x {texel, float output}
    for name, body in kernels:
        lines = body.split("\n")
        dest = set()
        for i, l in enumerate(lines):
            m = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off offset:(1024|2048)\s*$", l)
            if m:
                dest.update(range(int(m.group(1)), int(m.group(2)) + 1))
                m0 = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off\s*$", lines[i - 1]) if m.group(3) == "1024" else None
                if m0:
                    dest.update(range(int(m0.group(1)), int(m0.group(2)) + 1))
        assert len(dest) == 36, (name, sorted(dest))              # three banks of three digit fragments of four registers
        for l in lines:
            t = l.strip()
            if not t or t[0] in ";." or t.startswith(("v_mfma", "global_load_dwordx4")):
                continue
            regs = set()
            for m in re.finditer(r"v\[(\d+):(\d+)\]", t):
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bv(\d+)\b", t):
                regs.add(int(m.group(1)))
            assert not (regs & dest), (name, t)
        assert "scratch_" not in body and "flat_" not in body, name
        # round 6: between a step's a-operand reads and its MFMAs the only vector-memory wait is the hand-placed vmcnt(6).  (A compiler-visible
        # load left in flight across the loop's back edge -- the epilogue constants requested a round ahead were one -- makes the backend guard
        # the reuse of its register with a vmcnt(0) in front of EVERY step: a drain of the previous tile's result stores per step, +20 ... +80 %.)
        i = 0
        while i < len(lines):
            if "ds_read_b128" in lines[i]:
                j = i
                while j < len(lines) and "v_mfma" not in lines[j] and not lines[j].strip().startswith("s_barrier"):
                    if re.match(r"\s*s_waitcnt vmcnt", lines[j]):
                        assert ";;#ASMSTART" in lines[j - 1] and "vmcnt(6)" in lines[j], (name, j, lines[j])
                    j += 1
                i = j
            i += 1
        # the result stores take `uniform base + 32-bit lane offset`: no 64-bit vector address arithmetic next to them
        for k, l in enumerate(lines):
            if re.match(r"\s*global_store_(short|dword) ", l):
                assert re.search(r", s\[\d+:\d+\]\s*$", l), (name, l)
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) <= 256, (name, m and m.group(1))
