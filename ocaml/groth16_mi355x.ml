(* groth16_mi355x.ml -- seam 2: the body of src/groth16/groth16.ml with the prover on the MI355X.

   Install as src/groth16/groth16.ml; groth16.mli stays byte-identical (module Make(C : Curve.S) : Protocol.S with ...).
   The one change outside src/groth16: `module type G` of src/lib/zk/curve.ml / curve.mli gains
       val to_bytes : t -> bytes
       val of_bytes_exn : bytes -> t
   which every instance already has (they come with opam bls12-381's Fr, G1, G2, GT) -- a functor over an abstract
   Curve.S has no other way to hand group elements to a C library (ocaml/README.md).

   What is replaced (reference lines):
     prove   groth16.ml:235-237 + Base.prove :123-161 + sum_apply_powers :116-121 + QAP.eval (QAP.ml:120-135)
             -> one call zk_groth16_prove on a key uploaded once; r and s are drawn HERE, r first (:124-125)
     keygen  groth16.ml:227-233 + setup :45-108
             -> same exponents, all [x]_1 / [x]_2 in two fixed-base launches; registers the circuit with the library
     verify  groth16.ml:163-173 -> unchanged in substance: three pairings of the host's own Pairing
   The records and their yojson are the reference's (groth16.ml:24-43,110-114): the JSON of keys and proofs is the wire format. *)

open Zukelang
open Yojson_conv

module Make (C : Curve.S) = struct
  open C
  module Circuit = Circuit.Make (Fr)
  module QAP = QAP.Make (Fr)
  module Poly = QAP.Polynomial

  type f = Fr.t
  type circuit = Circuit.t
  type qap = QAP.t

  type pkey =
    { a : G1.t;
      d1 : G1.t;
      ti1 : G1.t list;
      ltd_mid : G1.t Var.Map.t;
      tiztd : G1.t list;
      b1 : G1.t;
      b2 : G2.t;
      d2 : G2.t;
      ti2 : G2.t list
    }
  [@@deriving yojson]

  type vkey =
    { one1 : G1.t;
      ltgm_io : G1.t Var.Map.t;
      one2 : G2.t;
      gm : G2.t;
      d : G2.t;
      ab : GT.t
    }
  [@@deriving yojson]

  type proof = { a : G1.t; b : G2.t; c : G1.t } [@@deriving yojson]

  (* ---------------------------------------------------------------- bytes *)

  let fr_bytes (xs : Fr.t list) = Mi355x.cat (List.map Fr.to_bytes xs)
  let g1_bytes (ps : G1.t list) = Mi355x.cat (List.map G1.to_bytes ps)
  let g2_bytes (ps : G2.t list) = Mi355x.cat (List.map G2.to_bytes ps)
  let g1_at b i = G1.of_bytes_exn (Bytes.sub b (96 * i) 96)
  let g2_at b i = G2.of_bytes_exn (Bytes.sub b (192 * i) 192)
  let values m = List.map snd (Var.Map.bindings m)

  (* ---------------------------------------------------------------- the circuit as three sparse matrices
     Rows = gates in Gate.Set.elements order (the ids QAP.build gives them, QAP.ml:22), columns = variables in Var.Map key
     order, entry = the coefficient QAP.build reads off the gate (QAP.ml:30-52). *)

  let index_of_vars (vars : Var.t list) : int Var.Map.t = Var.Map.of_list (List.mapi (fun i v -> (v, i)) vars)

  let matrices_of_gates (index : int Var.Map.t) (gates : Circuit.Gate.Set.t) =
    let rows sel =
      List.map
        (fun g -> List.map (fun (v, coeff) -> (Var.Map.find v index, Fr.to_bytes coeff)) (Var.Map.bindings (sel g)))
        (Circuit.Gate.Set.elements gates)
    in
    ( Mi355x.matrix_of_rows (rows (fun (g : Circuit.Gate.t) -> g.l)),
      Mi355x.matrix_of_rows (rows (fun (g : Circuit.Gate.t) -> g.r)),
      Mi355x.matrix_of_rows (rows (fun (g : Circuit.Gate.t) -> g.lhs)) )

  (* Without the circuit (a key read back from JSON, proved against a QAP): the coefficient of variable k in gate g is
     v_k(g) -- what QAP.decompile recovers (QAP.ml:96-118).  O(m n^2) field operations; only for sizes at which a dense
     QAP.t exists at all. *)
  let matrices_of_qap (index : int Var.Map.t) (qap : qap) n =
    let rows (polys : Poly.t Var.Map.t) =
      List.init n (fun g ->
          let x = Fr.of_int g in
          List.filter_map
            (fun (v, p) ->
              let coeff = Poly.apply p x in
              if Fr.(coeff = zero) then None else Some (Var.Map.find v index, Fr.to_bytes coeff))
            (Var.Map.bindings polys))
    in
    (Mi355x.matrix_of_rows (rows qap.v), Mi355x.matrix_of_rows (rows qap.w), Mi355x.matrix_of_rows (rows qap.y))

  (* ---------------------------------------------------------------- uploaded keys
     pkey is [@@deriving yojson], so the device handle cannot be a field of it.  It lives in a side table keyed by the
     (physically equal) pkey value; the entry dies with the key and the finaliser frees the device copy. *)

  module Handles = Ephemeron.K1.Make (struct
    type t = pkey

    let equal = ( == )
    let hash (k : pkey) = List.length k.ti1
  end)

  let handles : Unsigned.UInt64.t Handles.t = Handles.create 8

  (* set before the first prove of a key that will prove many times: the library then derives the key's Lagrange form on
     the device once (zk_groth16_pk_derive_lagrange; pays off after a few thousand proofs, header) *)
  let derive_lagrange_on_upload = ref false

  let upload (vars : Var.t list) (l, r, o) n (pkey : pkey) : Unsigned.UInt64.t =
    let var_at = Array.of_list vars in
    let m = Array.length var_at in
    let mid = Bytes.init m (fun i -> if Var.Map.mem var_at.(i) pkey.ltd_mid then '\001' else '\000') in
    (* declaration order of groth16.ml:24-34 per group, as include/zkmi355x.h lays it out *)
    let g1 = g1_bytes ((pkey.a :: pkey.d1 :: pkey.b1 :: pkey.ti1) @ pkey.tiztd @ values pkey.ltd_mid) in
    let g2 = g2_bytes (pkey.b2 :: pkey.d2 :: pkey.ti2) in
    let h = Mi355x.groth16_upload ~n ~m l r o ~mid ~g1 ~g2 in
    if !derive_lagrange_on_upload then Mi355x.(check (zk_groth16_pk_derive_lagrange h));
    Gc.finalise (fun _ -> ignore (Mi355x.zk_groth16_pk_free h)) pkey;
    Handles.replace handles pkey h;
    h

  let register_circuit (circuit : circuit) (qap : qap) (pkey : pkey) =
    let vars = List.map fst (Var.Map.bindings qap.v) in
    ignore (upload vars (matrices_of_gates (index_of_vars vars) circuit.Circuit.gates) (Poly.degree qap.QAP.target) pkey)

  let handle_of (qap : qap) (pkey : pkey) =
    match Handles.find_opt handles pkey with
    | Some h -> h
    | None ->
        let vars = List.map fst (Var.Map.bindings qap.v) in
        let n = Poly.degree qap.target in
        upload vars (matrices_of_qap (index_of_vars vars) qap n) n pkey

  (* ---------------------------------------------------------------- keygen *)

  let keygen rng (circuit : circuit) (qap : qap) : pkey * vkey =
    let n = Poly.degree qap.target in
    let alpha = Fr.gen rng in
    let beta = Fr.gen rng in
    let gamma = Fr.gen rng in
    let delta = Fr.gen rng in
    let tau = Fr.gen rng in
    let at_tau p = Poly.apply p tau in
    (* L_k(tau) = beta v_k(tau) + alpha w_k(tau) + y_k(tau) *)
    let l_tau =
      Var.Map.mapi
        (fun k vk -> Fr.((beta * at_tau vk) + (alpha * at_tau (Var.Map.find k qap.w)) + at_tau (Var.Map.find k qap.y)))
        qap.v
    in
    let io = Var.Set.union circuit.Circuit.inputs_public circuit.Circuit.outputs in
    let over set divisor = Var.Map.map (fun x -> Fr.(x / divisor)) (Var.Map.restrict set l_tau) in
    let l_mid = over circuit.Circuit.mids delta and l_io = over io gamma in
    let tau_powers count =
      let rec go acc x i = if i = count then List.rev acc else go (x :: acc) Fr.(x * tau) (i + 1) in
      go [] Fr.one 0
    in
    let z_over_delta = Fr.(at_tau qap.target / delta) in
    let ti = tau_powers (n + 2) in
    let tiz = List.map (fun x -> Fr.(x * z_over_delta)) (tau_powers (n - 1)) in
    (* every G1 element of both keys in ONE fixed-base launch, every G2 element in another *)
    let e1 = (alpha :: delta :: beta :: ti) @ tiz @ values l_mid @ values l_io in
    let e2 = (beta :: delta :: gamma :: ti) in
    let p1 = Mi355x.of_fr_many ~g2:false (fr_bytes e1) and p2 = Mi355x.of_fr_many ~g2:true (fr_bytes e2) in
    let n_ti = n + 2 and n_tiz = max (n - 1) 0 and n_mid = Var.Map.cardinal l_mid in
    let take b at off count = List.init count (fun i -> at b (off + i)) in
    let rekey m points = Var.Map.of_list (List.map2 (fun (k, _) p -> (k, p)) (Var.Map.bindings m) points) in
    let pkey : pkey =
      { a = g1_at p1 0;
        d1 = g1_at p1 1;
        b1 = g1_at p1 2;
        ti1 = take p1 g1_at 3 n_ti;
        tiztd = take p1 g1_at (3 + n_ti) n_tiz;
        ltd_mid = rekey l_mid (take p1 g1_at (3 + n_ti + n_tiz) n_mid);
        b2 = g2_at p2 0;
        d2 = g2_at p2 1;
        ti2 = take p2 g2_at 3 n_ti
      }
    in
    let vkey : vkey =
      { one1 = G1.one;
        ltgm_io = rekey l_io (take p1 g1_at (3 + n_ti + n_tiz + n_mid) (Var.Map.cardinal l_io));
        one2 = G2.one;
        gm = g2_at p2 2;
        d = pkey.d2;
        ab = Pairing.pairing pkey.a pkey.b2
      }
    in
    register_circuit circuit qap pkey;
    (pkey, vkey)

  (* ---------------------------------------------------------------- prove *)

  let prove rng (qap : qap) (pkey : pkey) (sol : f Var.Map.t) : proof =
    let handle = handle_of qap pkey in
    let r = Fr.gen rng in
    let s = Fr.gen rng in
    (* the reference folds over Dom(sol) and looks every key up in the QAP (`#!`: assert false when absent, var.ml:71-78) *)
    if not (Var.Set.equal (Var.Map.domain sol) (Var.Map.domain qap.v)) then assert false;
    let out = Bytes.create 384 in
    Mi355x.(
      check
        (zk_groth16_prove handle
           (bytes_start (fr_bytes (values sol)))
           (bytes_start (Fr.to_bytes r))
           (bytes_start (Fr.to_bytes s))
           (bytes_start out)));
    { a = g1_at out 0; b = G2.of_bytes_exn (Bytes.sub out 96 192); c = G1.of_bytes_exn (Bytes.sub out 288 96) }

  (* Throughput form: several proofs of one key in flight (header, zk_groth16_prove_async).  `jobs` = the (r, s) pairs,
     drawn by the caller in the reference's order; the witness is uploaded once. *)
  let prove_many (qap : qap) (pkey : pkey) (sol : f Var.Map.t) (jobs : (Fr.t * Fr.t) list) : proof list =
    let handle = handle_of qap pkey in
    let slots = 14 in
    Mi355x.(check (zk_groth16_reserve_slots handle (u32 slots)));
    Mi355x.(check (zk_groth16_set_witness handle (bytes_start (fr_bytes (values sol)))));
    let collect slot =
      let out = Bytes.create 384 in
      Mi355x.(check (zk_groth16_prove_wait handle (u32 slot) (bytes_start out)));
      ({ a = g1_at out 0; b = G2.of_bytes_exn (Bytes.sub out 96 192); c = G1.of_bytes_exn (Bytes.sub out 288 96) } : proof)
    in
    let rec go i pending acc = function
      | [] -> List.rev_append acc (List.map collect (List.rev pending))
      | (r, s) :: rest ->
          let slot = i mod slots in
          let acc, pending =
            if List.length pending = slots then
              match List.rev pending with
              | oldest :: others -> (collect oldest :: acc, List.rev others)
              | [] -> (acc, pending)
            else (acc, pending)
          in
          Mi355x.(
            check
              (zk_groth16_prove_async handle null_bytes (bytes_start (Fr.to_bytes r)) (bytes_start (Fr.to_bytes s)) (u32 slot)));
          go (i + 1) (slot :: pending) acc rest
    in
    go 0 [] [] jobs

  (* ---------------------------------------------------------------- verify
     e(A, B) = ab + e(sum_k w_k [L_k(tau)/gamma]_1, gamma) + e(C, delta), GT written additively as Curve.G has it.  The sum
     runs over the public coefficients; domains must agree (G.dot). *)
  let verify (input_output : f Var.Map.t) (vkey : vkey) (proof : proof) : bool =
    let e = Pairing.pairing in
    let public_part = G1.dot vkey.ltgm_io input_output in
    GT.(e proof.a proof.b - e public_part vkey.gm - e proof.c vkey.d = vkey.ab)
end
