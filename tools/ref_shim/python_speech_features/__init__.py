"""Import shim: python_speech_features is not installed in the survey container.
tools/gen_golden_segmenter.py only runs the reference's cut search
(simple_endpointing.py:21-79) on given energy curves; logfbank itself is NOT
provided here (its algorithm is restated in speechcatcher_amd/segmenter.py,
parity unpinned)."""


def logfbank(*args, **kwargs):
    raise NotImplementedError("shim: python_speech_features is not available")
