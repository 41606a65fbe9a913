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
rm -rf $W
