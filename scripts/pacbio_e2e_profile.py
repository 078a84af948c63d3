"""Where the time of the multi-locus PacBio route goes (vntr_finder.genotype_pacbio_loci): stage times of one pipelined run and a
cProfile of the extraction stage alone.   python scripts/pacbio_e2e_profile.py [n_loci]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
e.build()
from advntr_amd import workloads, vntr_finder, settings, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 896
loci, read_lists = workloads.make_pacbio_whole_reads(n)
_lib.require_gpu()
settings.MAX_ERROR_RATE = 0.3
vntr_finder.genotype_pacbio_loci(loci[:8], read_lists[:8], chunks=2)
for chunks in (4, 8, 16):
    T = {}
    t0 = time.perf_counter()
    vntr_finder.genotype_pacbio_loci(loci, read_lists, timings=T, chunks=chunks)
    print("chunks", chunks, "total %.3f" % (time.perf_counter() - t0), {k: round(v, 3) for k, v in T.items()})
pr = cProfile.Profile()
pr.enable()
ext = vntr_finder.extract_spanning_reads_multi([(l[0], l[1]) for l in loci], read_lists)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
