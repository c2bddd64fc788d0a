"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<name>.json.
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 reports half of a wide streaming read, MI355X_MICROARCH.md
section HBM) + WRITE_SIZE; both counters are in KiB."""
import csv, collections, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_sources_sha16():            # the digest bench.py recomputes before it quotes these figures (bench.py KERNEL_SOURCES)
    h = hashlib.sha256()
    for name in ["msm.hpp", "msm_impl.hpp", "bn254.hpp", "fips_asm.hpp"]:
        h.update(open(os.path.join(ROOT, "ethsnarks_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]
def agg(path):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r'zk::', '', r['Kernel_Name']); n = re.sub(r'\(.*', '', n).replace('void ', '')
        n = n.replace('Curve<Field<FqParams> >', 'G1').replace('Curve<Fq2>', 'G2').replace(' >', '>')
        a[n].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in a.items()}
fetch, write, out = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
res = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    res[k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --inflight 1 --no-extras --no-cpu-baseline, 2^20",
           "commit": os.environ.get("ZK_COMMIT", "unknown"), "kernel_sources_sha16": kernel_sources_sha16(),
           "config": {"workload": "chain", "logm": 20, "shards": 1},
           "note": "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE.  The x2 FETCH correction is documented for wide coalesced streams "
                   "(MI355X_MICROARCH.md, HBM); the accumulation kernels read 64-byte (G1) / 128-byte (G2) gathered points, for which it is "
                   "uncalibrated: read their figures as an UPPER bound (between 1x and 2x the raw counter).",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print("%-36s %10.1f MB" % (k, v["hbm_bytes_per_launch"] / 1e6))
