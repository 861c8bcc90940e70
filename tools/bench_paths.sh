#!/bin/bash
# Secondary paths, one call (runs on the GPU box): any-ratio rescale (1 and 8 frames per launch),
# RGBA16F target, pass 2 alone from both intermediate formats.
cd "${GRAFT_REPO_ROOT:-.}"
B="python tools/bench_scaled.py"
$B --path scaled --frames-per-launch 1
$B --path scaled --frames-per-launch 8
$B --path scaled --width 7680 --height 4320 --out-width 3840 --out-height 2160 --ring 16 --frames-per-launch 1
$B --path scaled --width 1920 --height 1080 --out-width 1366 --out-height 768 --ring 128 --frames-per-launch 16
# enlarging (round 6: the wave decodes each source pixel once): the renderer's own case, a 1080p clip in a larger view
$B --path scaled --width 1920 --height 1080 --out-width 3840 --out-height 2160 --frames-per-launch 8
$B --path scaled --width 1920 --height 1080 --out-width 3840 --out-height 2160 --frames-per-launch 1
$B --path scaled --width 1920 --height 1080 --out-width 2560 --out-height 1440 --frames-per-launch 8
$B --path rgba16f --frames-per-launch 1
$B --path rgba16f --frames-per-launch 16
$B --path render8
$B --path render8 --frames-per-launch 16
$B --path render16
$B --path render16 --frames-per-launch 16
