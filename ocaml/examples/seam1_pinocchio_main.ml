(* The reference's Pinocchio test executable (src/pinocchio/test/main.ml) over the GPU-backed curve instance (seam 1). *)
open Zk

module C = Bls12_381_mi355x
module F = Curve.Bls12_381.Fr
module Pinocchio = Pinocchio.Make (C)
module Test = Test.Make_suites (F) (Pinocchio.ZK)
