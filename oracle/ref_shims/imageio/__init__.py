"""Stand-in for imageio (video_source_file.py:8); image loading is done with PIL in the golden script."""
