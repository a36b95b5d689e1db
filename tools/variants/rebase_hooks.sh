#!/bin/bash
# Carry tools/variants/csrc_hooks.patch over a change of the product sources:
#   tools/variants/rebase_hooks.sh [<git rev the patch still applies to, default HEAD>]
# old product (git) + hooks -> instrumented tree; the product's change (old -> working tree) applied on top of it;
# new patch = diff working tree -> that.  Fails loudly when the product's change collides with a hook.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
mkdir -p $T/old/rtl-ws_amd $T/new/rtl-ws_amd $T/ins/rtl-ws_amd
(cd $ROOT && git archive $REV rtl-ws_amd/csrc tools/variants/csrc_hooks.patch | tar -x -C $T/old)
cp -r $T/old/rtl-ws_amd/csrc $T/ins/rtl-ws_amd/csrc
(cd $T/ins/rtl-ws_amd/csrc && patch -s -p3 < $T/old/tools/variants/csrc_hooks.patch)
cp -r $ROOT/rtl-ws_amd/csrc $T/new/rtl-ws_amd/csrc
(cd $T && diff -u -r -N old/rtl-ws_amd/csrc new/rtl-ws_amd/csrc > product.diff || true)
if [ -s $T/product.diff ]; then (cd $T/ins/rtl-ws_amd/csrc && patch -s -p3 -F3 < $T/product.diff); fi
find $T/ins -name "*.rej" | grep . && { echo "the product's change collides with a hook: rejects left in $T"; exit 1; }
find $T/ins -name "*.orig" -delete
rm -rf $T/pa $T/pb && mv $T/new $T/pa && mv $T/ins $T/pb
(cd $T && diff -u -r -N pa/rtl-ws_amd/csrc pb/rtl-ws_amd/csrc > $ROOT/tools/variants/csrc_hooks.patch || true)
rm -rf $T
grep -c "^diff" $ROOT/tools/variants/csrc_hooks.patch
