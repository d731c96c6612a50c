"""Development aid: which kernels of two builds of the library are the SAME machine code?

    python tools/kernel_isa_diff.py xrslam_amd/lib/libxrslam_hip_hold.so xrslam_amd/lib/libxrslam_hip.so

Extracts the gfx950 code objects of both shared objects (llvm-objdump --offloading, in a scratch directory), disassembles them and
compares every kernel's instruction stream (addresses stripped; branch targets are function-relative).  Used when a commit changes a
few kernels of a header all kernels share: the kernel revision (sha1 of *.hip.h) changes for the whole library, the per-launch counter
figures under profiles/ stay valid for the kernels this tool reports as identical."""
import collections
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

OD = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(lib):
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OD, "--offloading", so], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([OD, "-d", "--no-leading-addr", "--no-show-raw-insn", "-C", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            name, body = None, []
            for ln in txt.splitlines():
                m = re.match(r"^(?:[0-9a-f]+ )?<(.+)>:$", ln)
                if m:
                    if name:
                        out[name] = body
                    name, body = m.group(1), []
                elif name and ln.strip():
                    body.append(re.sub(r"\s*//.*$", "", ln).strip())
            if name:
                out[name] = body
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(name):
    return re.sub(r"^void ", "", name.split("(")[0]).replace("xrhip::", "")


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same, diff = [], []
    for k in sorted(set(a) | set(b)):
        ha = hashlib.sha1("\n".join(a.get(k, [])).encode()).hexdigest()[:10] if k in a else "absent"
        hb = hashlib.sha1("\n".join(b.get(k, [])).encode()).hexdigest()[:10] if k in b else "absent"
        (same if ha == hb else diff).append((short(k), len(a.get(k, [])), len(b.get(k, [])), ha, hb))
    print("identical instruction streams (%d):" % len(same))
    for k, na, nb, ha, _ in same:
        print("  %-44s %6d instructions  %s" % (k, na, ha))
    print("different (%d):" % len(diff))
    for k, na, nb, ha, hb in diff:
        print("  %-44s %6d -> %6d instructions  %s -> %s" % (k, na, nb, ha, hb))


if __name__ == "__main__":
    main()
