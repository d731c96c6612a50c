"""Headless harness around the C ABI: synthetic EuRoC-like sequences, the player's call sequence
(reference xrslam-pc/player/src/main.cpp:116-169) and trajectory evaluation.  Plumbing only."""
