#!/bin/bash
# smoke of the CLI's error paths: every command must end with a message and a non-zero status, never a signal
B=$GRAFT_REPO_ROOT/metakssd_amd/bin/metakssd
W=$(mktemp -d)
cd $W
$B shuffle -k 7 -s 4 -l 1 --seed 7 -o L1K7 > /dev/null
printf "" > empty.fq
printf "@r\nACGT\n" > trunc.fq
printf ">x\n" > hdr_only.fa
printf "@r\nACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIII\n" > one.fq
run() { timeout 30 "$@" > out.txt 2> err.txt < /dev/null; rc=$?; printf "%-70s rc=%d  %s\n" "$(echo "$@" | sed "s|$B|metakssd|" | cut -c1-70)" $rc "$(tail -c 120 err.txt | tr '\n' ' ')"; if [ $rc -ge 128 ]; then echo "   ^^^ SIGNAL"; fi; }
run $B
run $B dist
run $B dist -L nothere.shuf one.fq
run $B dist -L L1K7.shuf nothere.fq
run $B dist -L L1K7.shuf -A -o o1 empty.fq
run $B dist -L L1K7.shuf -A -o o2 trunc.fq
run $B dist -L L1K7.shuf -o o3 hdr_only.fa
run $B dist -L L1K7.shuf -A -o o4 one.fq
run $B dist -L L1K7.shuf -n 3 -Q 40 -o o5 one.fq
run $B dist -L L1K7.shuf -A --device 9 -o o6 one.fq
run $B dist -L L1K7.shuf -r ref one.fq
run $B set -u nothere
run $B set -u -o p1 o4
run $B set -i nothere -o p2 o4
run $B set -g nothere.tsv -o p3 o4
run $B set -x o4
run $B composite -r o4
run $B composite -r o4 -q o5
run $B composite -r o5 -q o4
run $B reverse o4
ls o1 o2 o4 2>/dev/null | tr '\n' ' '; echo
# stage II / search (none of these may start writing a 32 GiB index)
mkdir fakemco badmco
python3 - <<'PY'
import struct
st = open("o4/cofiles.stat", "rb").read()
shuf_id, = struct.unpack_from("<I", st, 0)
k, dr, comp, n = struct.unpack_from("<iiii", st, 8)
open("fakemco/mcofiles.stat", "wb").write(struct.pack("<Iiiii", shuf_id, k, dr, comp, n) + st[32:])
open("badmco/mcofiles.stat", "wb").write(struct.pack("<Iiiii", shuf_id + 1, k, dr, comp, n) + st[32:])
open("fakemco/mco.0", "wb").write(b"")
open("fakemco/mco.index.0", "wb").write(b"\0" * 4096)
PY
run $B dist -r nothere -o s1 o4
run $B dist -r badmco -o s2 o4
run $B dist -r fakemco -o s3 o4
run $B dist -r fakemco -o s4 nothere
run $B dist -r fakemco -o s5 fakemco
run $B dist -r fakemco -o s6 -M 7 o4
run $B dist -r fakemco -o s7 -f nothere.dat o4
run $B dist -r fakemco -o s8 -f one.fq -N 50 o4
run $B dist -o s9 o4 o5
cd / && rm -rf $W
