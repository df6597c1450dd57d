#!/bin/bash
set -o pipefail
bash profiles/collect.sh r05 $(cat profiles/r05/scripts/HEAD_COMMIT 2>/dev/null || echo HEAD)
