"""Import shim: the reference's simple_endpointing.py imports ffmpeg-python at
module level (speechcatcher/simple_endpointing.py:4) but the segment search that
the golden generator runs never touches it."""
