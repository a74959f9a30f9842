#!/bin/bash
# runs the reproducer over the wait sequences of a captured training step and its subsets (each in its own process)
cd "$(dirname "$0")"
run() { out=$(timeout -k 5 30 ./capture_topology "$1" 2>&1 | tail -1); echo "rc=$? | $1 | $out"; }
# M origin, S second pass chain, B helper of M, A helper of S, W weight-gradient stream
run "S<M M<S"
run "S<M W<S W<M M<S M<W"                                             # default training step
run "B<M M<B W<B W<M M<W"                                             # H beside G, one chain
run "S<M A<S S<A B<M M<B M<S"                                         # H beside G, two chains, no weight-gradient stream
run "S<M A<S S<A B<M M<B A<S W<A W<S S<A B<M W<B W<M M<B M<S M<W"     # the aborting step
run "S<M A<S W<A S<A M<S M<W"
run "S<M A<S W<A M<W M<A M<S"
run "S<M A<S W<A W<S S<A M<S M<W"
run "S<M A<S S<A W<A S<A M<S M<W"
run "S<M A<S S<A W<S W<A S<A M<S M<W"
run "S<M W<S A<S W<A S<A M<S M<W"
run "S<M A<S S<A A<S S<A M<S"
run "S<M A<S S<A A<S W<A S<A M<S M<W"
# workaround candidates: every helper forked from the origin first
run "S<M A<M B<M W<M A<S S<A B<M M<B A<S W<A W<S S<A B<M W<B W<M M<B M<S M<W M<A"
run "W<M S<M A<S S<A B<M M<B A<S W<A W<S S<A B<M W<B W<M M<B M<S M<W"
# the minimal reproducer: two non-origin streams waiting for each other
run "S<M A<S S<A M<S"
run "S<M A<S M<A M<S"
