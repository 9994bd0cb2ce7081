import os, sys
sys.path.insert(0, os.getcwd())
from concurrent.futures import ThreadPoolExecutor
import bench
from seeksv_amd import host, synth
cores = bench.effective_cpus()
os.environ["SSV_BGZF_LEVEL"] = "4"; os.environ.setdefault("SSV_WRITE_THREADS", str(cores))
w = synth.Workload(genome_frac=0.125, depth=30, n_sv=1250)
d = "/dev/shm/apitrace"; os.makedirs(d, exist_ok=True)
chunk = 500_000; starts = list(range(0, w.n_total, chunk))
def batches():
    with ThreadPoolExecutor(max_workers=cores) as ex:
        for i in range(0, len(starts), 2 * cores):
            yield from ex.map(lambda g: w.generate_host(g, min(chunk, w.n_total - g), False, True), starts[i:i + 2 * cores])
host.write_bam(d + "/s.bam", w.names, w.lens, batches())
with open(d + "/j.txt", "w") as f:
    for j in w.junctions:
        f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")
host.write_bam(d + "/e.bam", w.names, w.lens, [])
open(d + "/e.clip", "w").close()
