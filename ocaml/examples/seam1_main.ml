(* The reference's own Groth16 test executable (src/groth16/test/main.ml) over the GPU-backed curve instance: seam 1 only,
   no file of src/groth16 touched.  Build it beside that file (same dune stanza, `(libraries groth16 test)`). *)
open Zk

module C = Bls12_381_mi355x
module F = Curve.Bls12_381.Fr
module Groth16 = Groth16.Make (C)
module Test = Test.Make_suites (F) (Groth16)
