"""Per-kernel register / scratch / occupancy table from a hipcc -Rpass-analysis=kernel-resource-usage log.

    hipcc ... -Rpass-analysis=kernel-resource-usage -c glv_inst.hip 2> log.txt;  python tools/regs.py log.txt [STATEFUL ...]
"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = set(sys.argv[2:])
blocks = txt.split("remark: Function Name: ")[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"glv_frame_kernel<(.*)>", d)
    if not m:
        continue
    parts = m.group(1).replace(" ", "").split(",")
    # LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TWREG, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, STATEFUL, WPRE
    if want and parts[11] not in want:
        continue
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"nn=2^{parts[0]} in={parts[1]} log={parts[2]} slots={parts[3]} E=2^{parts[10]} stateful={parts[11]} vgpr={g('VGPRs')} agpr={g('AGPRs')} "
          f"scratch={scratch} occ={occ} lds={lds}")
