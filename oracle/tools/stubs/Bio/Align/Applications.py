class MuscleCommandline(object):
    def __init__(self, *a, **k):
        raise RuntimeError("muscle / biopython are not available in this image")
