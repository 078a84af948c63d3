"""Import-only placeholder for biopython (absent from this image).

TEST INFRASTRUCTURE, build container only.  advntr/profile_hmm.py:6-7 imports
MuscleCommandline and AlignIO at module level; the golden generator never calls
build_profile_hmm_for_repeats with more than one repeat un-aligned (muscle is not
in the image either), so neither name is ever executed.
"""
class AlignIO(object):
    @staticmethod
    def read(*a, **k):
        raise RuntimeError("biopython is not available in this image")
