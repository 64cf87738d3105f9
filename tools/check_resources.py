"""Register / scratch report of every kernel of libmmx_hip.so (hipcc -Rpass-analysis=kernel-resource-usage over csrc/*.hip) and
the build's spill gate: the kernels the engines' DEFAULTS launch must have ScratchSize 0 (`make -C minimax-speech_amd/csrc check`).

    python tools/check_resources.py [file.hip ...] [--all]      (--all: print every kernel, not only the gated / spilling ones)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minimax-speech_amd", "csrc")
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -std=c++20 -Wno-unused-result -Wno-pass-failed -Rpass-analysis=kernel-resource-usage".split()
# demangled-name prefixes of the kernels the default configurations launch (mmx/flow.py, mmx/llm.py, mmx/dac.py tile rules)
GATED = [
    "est_tail_kernel<unsigned short, 64, 2, 8, 1, 1, 2, false>",      # bf16 build, 64-row tiles
    "est_tail_kernel<unsigned short, 32, 8, 8, 1, 1, 2, false>",      # bf16 build, 32-row tiles
    "est_tail_kernel<unsigned short, 16, 8, 4, 1, 1, 4, false>",      # bf16 build, 16-row tiles
    "est_tail_kernel<unsigned short, 64, 2, 8, 2, 1, 2, false>",         # split build, 64-row tiles (K-halved attention tile)
    "est_tail_kernel<unsigned short, 32, 2, 8, 2, 1, 4, false>",         # split build
    "est_tail_kernel<unsigned short, 16, 4, 8, 2, 1, 4, false>",      # split build, 16-row tiles (8 waves)
    "est_tail_kernel<unsigned short, 64, 2, 8, 2, 1, 2, true>", "est_tail_kernel<unsigned short, 32, 2, 8, 2, 1, 4, true>",       # split build, weight planes
    "est_tail_kernel<unsigned short, 16, 4, 8, 2, 1, 4, true>",
    "est_resnet_kernel<unsigned short, 64, 2, 8, 1, false>", "est_resnet_kernel<unsigned short, 32, 4, 8, 1, false>",
    "est_resnet_kernel<unsigned short, 32, 2, 4, 2, false>", "est_resnet_kernel<unsigned short, 32, 4, 4, 2, false>",
    "est_resnet_kernel<unsigned short, 32, 4, 8, 2, false>", "est_resnet_kernel<unsigned short, 32, 4, 8, 2, true>",   # split build, 8 waves (cin = 256)
    "est_resnet_kernel<unsigned short, 32, 2, 4, 2, true>", "est_resnet_kernel<unsigned short, 32, 4, 4, 2, true>",
    "skinny3_kernel", "decode_attn_kernel", "sample_step_kernel", "attn_flash_kernel", "attn_flash_x_kernel", "attn_relpos",
    "dac_ru_kernel", "gemm_win_kernel",
]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def report(path):
    with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
        err = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", path, "-o", tmp.name], capture_output=True, text=True, cwd=CSRC).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|TotalSGPRs): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k.split(" ")[0] + ("_spill" if "Spill" in k else "")] = int(v)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(anonymous namespace\)::|^void ", "", n.split("(")[0] if "<" not in n else n[:n.rindex(">") + 1])
    return rows


def main():
    files = [a for a in sys.argv[1:] if a.endswith(".hip")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    show_all = "--all" in sys.argv
    bad = []
    for f in files:
        for r in report(os.path.join(CSRC, f)):
            gated = any(g in r["name"] for g in GATED)
            spills = r.get("ScratchSize", 0) > 0
            if show_all or spills or gated:
                print(f"{f:16s} {r['name'][:96]:96s} vgpr {r.get('VGPRs', 0):3d} agpr {r.get('AGPRs', 0):3d} scratch {r.get('ScratchSize', 0):4d} "
                      f"occ {r.get('Occupancy', 0)}{'  GATED' if gated else ''}{'  SPILLS' if spills else ''}")
            if gated and spills:
                bad.append(r["name"])
    if bad:
        print("\nspilling kernels on a default path:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print("\nno gated default-path kernel spills")


if __name__ == "__main__":
    main()
