"""FETCH_SIZE / WRITE_SIZE summaries of the default bench workload (tools/pmc_summary.py output) -> profiles/<tag>_pmc_kernels.json: HBM bytes per
launch of the detector's kernel families, which bench.py attaches as roofline.traffic while the detector's sources still hash the same.
usage: python tools/pmc_kernels_json.py <fetch.txt> <write.txt> <tag>
Bytes = FETCH_SIZE x 2 (MI355X_MICROARCH.md, section HBM: on gfx950 the counter reports half the bytes of wide coalesced reads) + WRITE_SIZE
as reported (the guide calls it uncalibrated; for fhog_split_ml_k it comes to 0.92 of the bytes the kernel is known to write), KB -> bytes.
A family's "launch" is what bench.py's HIP events bracket: one kernel for fhog / score, the 20 resize launches of a batch for pyramid,
score_screen_k + score_list_k for score_screened."""
import json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.search(r"^(.*?)\s+n=(\d+)\s+%s=([0-9.e+]+)" % counter, line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


def main():
    fetch, write, tag = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    def total_kb(table, key):
        return sum(n * kb for name, (n, kb) in table.items() if key in name)
    def launches(table, key):
        return sum(n for name, (n, kb) in table.items() if key in name)
    # batches of the pass that ran a family: the launches of the one kernel it launches once per batch (the pass holds bench.py's
    # dense-scoring leg as well: score_roll_k and score_screen_k each see a part of the batches, the pyramid and FHOG kernels all of them)
    fam = {"pyramid": (["resize_rows_k"], "fhog_split_ml_k"), "fhog": (["fhog_split_ml_k"], "fhog_split_ml_k"), "score": (["score_roll_k"], "score_roll_k"),
           "score_screened": (["score_screen_k", "score_list_k"], "score_screen_k")}
    kernels = {}
    for name, (keys, per_batch) in fam.items():
        batches = launches(fetch, per_batch)
        f = sum(total_kb(fetch, k) for k in keys); w = sum(total_kb(write, k) for k in keys)
        if batches == 0 or (f == 0 and w == 0):
            continue
        kernels[name] = {"kernels": keys, "batches_in_pass": batches, "fetch_size_kb_per_launch": f / batches, "write_size_kb_per_launch": w / batches,
                         "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 / batches}
    import bench
    d = {"detect_batch": 128, "frame": "1920x1080", "kernels": kernels, "detector_sha256_16": bench.detector_hash(),
         "source": "profiles/%s_pmc_fetch_size.txt + profiles/%s_pmc_write_size.txt (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 1`, "
                   "a pass each; per batch of the step; FETCH_SIZE x 2: the gfx950 correction of MI355X_MICROARCH.md section HBM; WRITE_SIZE as reported)" % (tag, tag)}
    for path in ("profiles/%s_pmc_kernels.json" % tag, "gpurun_out/%s/pmc_kernels.json" % tag):
        json.dump(d, open(path, "w"), indent=1)
    print(json.dumps(d)[:600])


main()
