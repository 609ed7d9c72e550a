(* bls12_381_mi355x.ml -- seam 1: a GPU-backed instance of Curve.S (src/lib/zk/curve.mli:46-54).

   Goes to src/lib/zk/.  `Groth16.Make(Bls12_381_mi355x)` / `Pinocchio.Make(Bls12_381_mi355x)` then run the reference's
   protocol code UNCHANGED with every `apply_powers`, `dot` and `powers` (curve.ml:91-118) executed as one multi-scalar
   product / fixed-base kernel on the MI355X.  The O(m n) structure of `sum_apply_powers` (groth16.ml:116-121) remains with
   this seam alone -- seam 2 (groth16_mi355x.ml, pinocchio_mi355x.ml) removes it.

   Fr, GT and Pairing are the host's own (opam bls12-381): a verifier needs a handful of pairings, not a GPU. *)

module Base = Curve.Bls12_381

module Make_group (G : sig
  include Curve.G with type fr := Base.Fr.t

  val to_bytes : t -> bytes
  val of_bytes_exn : bytes -> t
  val g2 : bool
  val point_bytes : int
end) =
struct
  include G

  let scalars_bytes (cs : Base.Fr.t list) = Mi355x.cat (List.map Bls12_381.Fr.to_bytes cs)
  let points_bytes (ps : t list) = Mi355x.cat (List.map to_bytes ps)

  let split (b : bytes) : t list =
    List.init (Bytes.length b / point_bytes) (fun i -> of_bytes_exn (Bytes.sub b (point_bytes * i) point_bytes))

  (* G.of_Fr over a list: one fixed-base launch (curve.ml:180 mapped) *)
  let of_Fr_many (ss : Base.Fr.t list) : t list = split (Mi355x.of_fr_many ~g2 (scalars_bytes ss))

  (* curve.ml:106-109: d + 1 points g^(s^0) .. g^(s^d) *)
  let powers d s = split (Mi355x.powers ~g2 d (Bls12_381.Fr.to_bytes s))

  (* curve.ml:112-118: sum_i c_i x_i; runs out of coefficients quietly, of points with Invalid_argument "apply_powers" *)
  let apply_powers (cs : Base.Fr.t Polynomial.t) (xis : t list) : t =
    match cs with
    | [] -> zero
    | _ -> of_bytes_exn (Mi355x.msm ~g2 (points_bytes xis) (scalars_bytes cs))

  (* curve.ml:94-103: equal key sets or `assert false` *)
  let dot (m : t Var.Map.t) (c : Base.Fr.t Var.Map.t) : t =
    if not (Var.Set.equal (Var.Map.domain m) (Var.Map.domain c)) then assert false;
    apply_powers (List.map snd (Var.Map.bindings c)) (List.map snd (Var.Map.bindings m))
end

module Fr = Base.Fr
module GT = Base.GT
module Pairing = Base.Pairing

(* Curve.Bls12_381 is sealed to Curve.S, which has no byte access; the type equalities it exports (curve.mli:56-60) let the
   opam module's own to_bytes / of_bytes_exn be used on its points. *)
module G1 = Make_group (struct
  include Base.G1

  let to_bytes = Bls12_381.G1.to_bytes
  let of_bytes_exn = Bls12_381.G1.of_bytes_exn
  let g2 = false
  let point_bytes = 96
end)

module G2 = Make_group (struct
  include Base.G2

  let to_bytes = Bls12_381.G2.to_bytes
  let of_bytes_exn = Bls12_381.G2.of_bytes_exn
  let g2 = true
  let point_bytes = 192
end)
