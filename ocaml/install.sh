#!/bin/sh
# Installs the MI355X binding into a zukelang checkout:  ocaml/install.sh /path/to/zukelang
# (needs opam packages ctypes, ctypes-foreign; libzkmi355x.so on the loader path or ZK_LIBZKMI355X_PATH set).
# What it changes is listed in ocaml/README.md; `git diff` in the checkout shows it all.
set -e
Z=${1:?usage: install.sh /path/to/zukelang}
HERE=$(cd "$(dirname "$0")" && pwd)
# new modules of library `zk`
cp "$HERE/mi355x.ml" "$HERE/bls12_381_mi355x.ml" "$HERE/r1cs_file.ml" "$Z/src/lib/zk/"
# seam 2: the two protocol bodies; their .mli files stay as they are
cp "$HERE/groth16_mi355x.ml" "$Z/src/groth16/groth16.ml"
cp "$HERE/pinocchio_mi355x.ml" "$Z/src/pinocchio/pinocchio.ml"
# library `zk` links ctypes
sed -i 's/(libraries zarith bls12-381 /(libraries ctypes ctypes.foreign zarith bls12-381 /' "$Z/src/lib/zk/dune"
# module type G gains byte access (every instance already has both functions)
for f in curve.ml curve.mli; do
  sed -i '0,/^  val pp : t printer$/s//  val pp : t printer\n  val to_bytes : t -> bytes\n  val of_bytes_exn : bytes -> t/' "$Z/src/lib/zk/$f"
done
# the seam-1 executables beside the reference's own test mains
cp "$HERE/examples/seam1_main.ml" "$Z/src/groth16/test/seam1_main.ml"
cp "$HERE/examples/seam1_pinocchio_main.ml" "$Z/src/pinocchio/test/seam1_main.ml"
sed -i 's/(names main)/(names main seam1_main)/' "$Z/src/groth16/test/dune" "$Z/src/pinocchio/test/dune"
echo "installed; now: (cd $Z && dune build && dune runtest)"
