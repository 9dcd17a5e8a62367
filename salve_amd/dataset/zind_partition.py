"""Official ZInD train / val / test building split (1575 tours), as salve/dataset/zind_partition.py carries it.

The ids are data (https://github.com/zillow/zind `zind_partition.json`); they live in `zind_partition.json` next to
this file, written by tests/golden/make_golden.py from the imported reference.
"""

import json
from pathlib import Path

with open(Path(__file__).resolve().parent / "zind_partition.json") as _f:
    DATASET_SPLITS = json.load(_f)
