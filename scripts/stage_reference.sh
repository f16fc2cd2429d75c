#!/bin/bash
# Stage the read-only reference checkout into the git-ignored oracle/_ref/ so that it rides the gpurun
# snapshot to the GPU box (which has no /root/reference).  TEST INFRASTRUCTURE: the staged copy is the
# checker — tests/test_gpu_reference_loop.py runs its UNMODIFIED attack.py with `--gar native-*` next to
# the reference's own rules, tests/test_gpu_full_size_o1.py and bench.py's cpu_baseline leg import its
# aggregators as the CPU reference.  Nothing is copied into the tracked tree (oracle/_ref/ is listed in
# .gitignore, not in .gpurunignore); the product never imports it.
#
#   scripts/stage_reference.sh [SRC=/root/reference] [DST=<repo>/oracle/_ref/reference]
set -euo pipefail
here="$(cd "$(dirname "$0")/.." && pwd)"
src="${1:-${BM_REFERENCE_DIR:-/root/reference}}"
dst="${2:-$here/oracle/_ref/reference}"
if [ ! -f "$src/aggregators/__init__.py" ]; then
  echo "stage_reference: no reference checkout at $src (nothing staged)" >&2
  exit 3
fi
rm -rf "$dst"
mkdir -p "$dst"
# sources only: no bytecode, no VCS data, no dangling submodule link (experiments/models/wide_resnet.py)
(cd "$src" && find . -type f \( -name '*.py' -o -name 'LICENSE' -o -name 'README.md' \) ! -path './.git/*' -print0 \
  | tar --null -T - -cf -) | tar -xf - -C "$dst"
mkdir -p "$dst/experiments/datasets/cache"
(cd "$src" && find . -type f -name '*.py' ! -path './.git/*' -print0 | sort -z | xargs -0 sha256sum) > "$dst/../MANIFEST.sha256"
echo "staged $(find "$dst" -name '*.py' | wc -l) reference files into $dst"
