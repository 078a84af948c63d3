"""The genotype CLI at a larger scale than the tests use: N loci in a sqlite model database, 30x-like read sets (150-base
reads from two alleles per locus + background), prefilter -> scoring -> genotypes; reports the wall time and how many
planted genotypes come back.  Usage: python scripts/cli_scale_check.py [n_loci]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from advntr_amd import models, workloads, vntr_finder

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(17)
tmp = tempfile.mkdtemp()
db = os.path.join(tmp, "m.db")
models.create_vntrs_database(db)
reads, truth = [], {}
for k in range(n_loci):
    plen = int(rng.integers(8, 40))
    pattern, left, right = workloads.rand_seq(rng, plen), workloads.rand_seq(rng, 500), workloads.rand_seq(rng, 500)
    ref_copies = int(rng.integers(2, 6))
    v = models.ReferenceVNTR(1000 + k, pattern, 5000 * k, "chr%d" % (1 + k % 22), None, None, ref_copies)
    v.init_from_xml([pattern] * ref_copies, left, right)
    models.save_reference_vntr_to_database(v, db)
    max_c = max(1, (150 - 40) // plen)                       # alleles a 150-base read can still span
    alleles = sorted(int(x) for x in rng.integers(1, max_c + 1, 2))
    truth[1000 + k] = alleles
    for c in alleles:
        allele = left + pattern * c + right
        for _ in range(30):
            st = int(rng.integers(500 - 130, 500 - 20))
            s = allele[st:st + 150]
            reads.append(s if rng.random() < 0.5 else vntr_finder.reverse_complement(s))
reads += [workloads.rand_seq(rng, 150) for _ in range(20 * n_loci)]
order = rng.permutation(len(reads))
fa = os.path.join(tmp, "reads.fa")
with open(fa, "w") as fh:
    for i in order:
        fh.write(">r%d\n%s\n" % (i, reads[i]))
t0 = time.perf_counter()
out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--models", db, "--reads", fa, "--prefilter-both-strands"], cwd=ROOT,
                     stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
dt = time.perf_counter() - t0
got = {int(out[i]): out[i + 1] for i in range(0, len(out) - 1, 2)}
ok = sum(got.get(v) == "/".join(str(a) for a in al) for v, al in truth.items())
print("loci %d, reads %d: CLI wall time %.1f s, planted genotypes recovered %d/%d" % (n_loci, len(reads), dt, ok, n_loci))
