"""Empty stand-in so that `import ffmpeg` (video_source_file.py:12) succeeds; no video I/O is used."""
