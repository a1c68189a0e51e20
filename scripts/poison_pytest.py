"""Run pytest with torch's caching allocator pre-loaded with NaN-filled blocks: every later torch.empty() hands out NaNs
instead of whatever the fresh box happened to have in memory (usually zeros), so a kernel that reads a buffer element it
was supposed to be given initialised -- or that nobody wrote -- fails loudly instead of once in a few cold starts.
Usage: python scripts/poison_pytest.py <pytest args>"""
import sys

import pytest
import torch

if torch.cuda.is_available():
    blocks = []
    for shift in (28, 26, 24, 22, 20, 18, 16, 14, 12, 10):
        for _ in range(6 if shift < 26 else 2):
            blocks.append(torch.full((1 << shift,), float('nan'), dtype=torch.float32, device='cuda'))
    torch.cuda.synchronize()
    del blocks
sys.exit(pytest.main(sys.argv[1:]))
