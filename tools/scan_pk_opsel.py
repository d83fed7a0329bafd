"""Build-time audit (csrc/common.h, "packed-fp32 guard").  On gfx950 a packed fp32 VALU op (v_pk_mul/add/fma_f32) that takes the HIGH
register of a VGPR pair for its LOW lane (op_sel bit = 1 on that source) reads zero for that operand in lanes 48-63, some of the time, when the
pair was written by a global_load_dwordx2 and another wave of the SIMD issues MFMAs (reproduced in isolation by tools/probes/pk_mfma.hip,
profiles/r05_defect_isa/README.md; round 3 met it as two wrong-result defects).  The form - and every other packed fp32 op - is kept out of
the device code (-fno-slp-vectorize, -packed-fp32-ops); this script compiles every csrc/*.hip to gfx950 assembly with the flags of build.sh
and lists every packed fp32 op that still has the form (and, as a second class, those whose selected register was last written by a
vector-memory load).  Exit status 1 if any is found.      python tools/scan_pk_opsel.py [file.s ...]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = re.compile(r'^\s*v_pk_(mul|add|fma)_f32\s+(.*)$')
DST = re.compile(r'^\s*([a-z_0-9]+)\s+(v\[\d+:\d+\]|v\d+)(?=[\s,]|$)')


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return [int(m.group(1))] if m else []


def scan(path):
    hits, kernel, last = {}, "?", {}
    for line in open(path):
        if line.startswith("_Z") and ":" in line:
            kernel, last = line.split(":")[0], {}
            continue
        code = line.split(";")[0]
        m = PK.match(code)
        if m:
            body = m.group(2)
            sel = re.search(r'op_sel:\[([01,]+)\]', body)
            ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', re.sub(r'\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp).*$', '', body))]
            if sel:
                bits = sel.group(1).split(",")
                for k, b in enumerate(bits):
                    if b == "1" and k + 1 < len(ops):
                        r = regs(ops[k + 1])
                        if len(r) == 2:
                            hits.setdefault(kernel, []).append(("[after a memory load] " if last.get(r[1]) == "vmem" else "") + code.strip())
        d = DST.match(code)
        if d:
            mnem = d.group(1)
            kind = "vmem" if re.match(r'(global|buffer|flat|scratch)_load', mnem) else ("lds" if mnem.startswith("ds_") else "valu")
            if re.match(r'(global|buffer|flat|scratch)_(store|atomic)', mnem) or mnem.startswith("ds_write") or mnem.startswith("ds_store"):
                continue
            for r in regs(d.group(2)):
                last[r] = kind
    return hits


def build_flags():
    """the code-generation flags of csrc/build.sh that are not in the fixed command line below (e.g. -fno-slp-vectorize, -Xclang -target-feature -Xclang -packed-fp32-ops)"""
    text = open(os.path.join(ROOT, "manipose_amd", "csrc", "build.sh")).read()
    m = re.search(r'FLAGS="([^"]*)"', text)
    toks, out, i = (m.group(1).split() if m else []), [], 0
    while i < len(toks):
        if toks[i] == "-Xclang" and i + 1 < len(toks):          # -Xclang <arg> pairs (e.g. -target-feature -packed-fp32-ops)
            out += toks[i:i + 2]
            i += 2
            continue
        if toks[i].startswith("-f") and toks[i] != "-fPIC":
            out.append(toks[i])
        i += 1
    return out


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="pkscan_")
        procs = []
        for src in sorted(glob.glob(os.path.join(ROOT, "manipose_amd", "csrc", "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            procs.append((src, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + build_flags() + ["-S", "--cuda-device-only", "-o", out, src],
                                                     stderr=subprocess.PIPE, text=True)))
        failed = []
        for src, out, p in procs:
            err = p.communicate()[1]
            if p.returncode == 0 and os.path.getsize(out) > 0:
                files.append(out)
            else:
                failed.append(src)
                sys.stderr.write(f"scan_pk_opsel: compiling {src} failed (exit {p.returncode}):\n{err[-2000:]}\n")
        # an audit that skipped a source has audited nothing: every csrc/*.hip must have compiled
        if failed or len(files) != len(procs):
            print(f"{len(failed)} of {len(procs)} sources did not compile: {[os.path.basename(f) for f in failed]}")
            sys.exit(2)
    total, packed = 0, 0
    for f in files:
        packed += sum(1 for line in open(f) if PK.match(line.split(";")[0]))
    for f in files:
        for k, v in scan(f).items():
            total += len(v)
            print(f"{os.path.basename(f)}: {k[:100]}: {len(v)} packed fp32 ops take the high register of a pair for their low lane, e.g. {v[0][:130]}")
    # (with build.sh's -target-feature -packed-fp32-ops no v_pk_*_f32 exists at all, so the op_sel form cannot either: both counts are printed)
    print(f"{packed} packed fp32 ops (v_pk_mul/add/fma_f32) in the generated code")
    print(f"{total} suspicious packed ops in {len(files)} files")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
