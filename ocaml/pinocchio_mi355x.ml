(* pinocchio_mi355x.ml -- seam 2 for Pinocchio Protocol 2: the body of src/pinocchio/pinocchio.ml with the prover on the MI355X.

   Install as src/pinocchio/pinocchio.ml; pinocchio.mli stays byte-identical (Make(C).NonZK / ZK : Protocol.S).  Needs the two
   `val`s in `Curve.G` described in groth16_mi355x.ml.

   What is replaced (reference lines):
     ZK.prove     pinocchio.ml:559-561 + ZKCompute.f :427-514 + QAP.eval  -> zk_pinocchio_prove; dv, dw, dy drawn HERE in the
                  order of :428-430
     NonZK.prove  pinocchio.ml:536-538 + Compute.f :210-248                -> the same call with dv = dw = dy = 0 (no rng use)
     keygen       pinocchio.ml:530-534 + KeyGen.generate :77-189           -> same exponents (rv, rw, s, av, aw, ay, b, gm drawn
                  in the order of :83-91), every key point in two fixed-base launches
     verify       pinocchio.ml:540-541,563 + Verify.f :254-420             -> zk_pinocchio_verify (13 pairings on the host, in the
                  library: it takes only G1 / G2 points, so no GT encoding is involved).  Where the reference `assert`s the
                  four knowledge-of-coefficient checks and returns the divisibility check, this returns false for any failing
                  check.
   Records and their yojson are the reference's (pinocchio.ml:37-75,195-208). *)

open Zukelang
open Yojson_conv

module Make (C : Curve.S) = struct
  open C
  module Circuit = Circuit.Make (C.Fr)
  module QAP = QAP.Make (C.Fr)
  module Poly = QAP.Polynomial

  type circuit = Circuit.t
  type qap = QAP.t

  type pkey =
    { vv : G1.t Var.Map.t;
      ww : G2.t Var.Map.t;
      yy : G1.t Var.Map.t;
      vav : G1.t Var.Map.t;
      waw : G2.t Var.Map.t;
      yay : G1.t Var.Map.t;
      si : G1.t list;
      bvwy : G1.t Var.Map.t;
      si2 : G2.t list;
      vt : G1.t;
      wt : G2.t;
      yt : G1.t;
      vavt : G1.t;
      wawt : G2.t;
      yayt : G1.t;
      vbt : G1.t;
      wbt : G1.t;
      ybt : G1.t;
      v_all : G1.t Var.Map.t;
      w_all : G1.t Var.Map.t
    }
  [@@deriving yojson]

  type vkey =
    { one : G1.t;
      one2 : G2.t;
      av : G2.t;
      aw : G1.t;
      ay : G2.t;
      gm2 : G2.t;
      bgm : G1.t;
      bgm2 : G2.t;
      yt : G2.t;
      vv_io : G1.t Var.Map.t;
      ww_io : G2.t Var.Map.t;
      yy_io : G1.t Var.Map.t
    }
  [@@deriving yojson]

  type proof =
    { vv : G1.t;
      ww : G2.t;
      yy : G1.t;
      h : G1.t;
      vavv : G1.t;
      waww : G2.t;
      yayy : G1.t;
      bvwy : G1.t
    }
  [@@deriving yojson]

  let fr_bytes (xs : Fr.t list) = Mi355x.cat (List.map Fr.to_bytes xs)
  let g1_bytes (ps : G1.t list) = Mi355x.cat (List.map G1.to_bytes ps)
  let g2_bytes (ps : G2.t list) = Mi355x.cat (List.map G2.to_bytes ps)
  let g1_at b i = G1.of_bytes_exn (Bytes.sub b (96 * i) 96)
  let g2_at b i = G2.of_bytes_exn (Bytes.sub b (192 * i) 192)
  let values m = List.map snd (Var.Map.bindings m)
  let keys m = List.map fst (Var.Map.bindings m)

  (* ---------------------------------------------------------------- circuit rows (as in groth16_mi355x.ml) *)

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
    (Mi355x.matrix_of_rows (rows qap.QAP.v), Mi355x.matrix_of_rows (rows qap.QAP.w), Mi355x.matrix_of_rows (rows qap.QAP.y))

  (* ---------------------------------------------------------------- uploaded keys: pools in the order of include/zkmi355x.h
       g1: vv | yy | vav | yay | bvwy (I_mid each) | si (n+1) | v_all (m) | w_all (m) | vt | yt | vavt | yayt | vbt | wbt | ybt
       g2: ww | waw (I_mid each) | si2 (n+1) | wt | wawt *)

  module Handles = Ephemeron.K1.Make (struct
    type t = pkey

    let equal = ( == )
    let hash (k : pkey) = List.length k.si
  end)

  let handles : Unsigned.UInt64.t Handles.t = Handles.create 8
  let derive_lagrange_on_upload = ref false

  let upload (vars : Var.t list) (l, r, o) n (k : pkey) : Unsigned.UInt64.t =
    let var_at = Array.of_list vars in
    let m = Array.length var_at in
    let mid = Bytes.init m (fun i -> if Var.Map.mem var_at.(i) k.vv then '\001' else '\000') in
    let g1 =
      g1_bytes
        (values k.vv @ values k.yy @ values k.vav @ values k.yay @ values k.bvwy @ k.si @ values k.v_all @ values k.w_all
        @ [ k.vt; k.yt; k.vavt; k.yayt; k.vbt; k.wbt; k.ybt ])
    in
    let g2 = g2_bytes (values k.ww @ values k.waw @ k.si2 @ [ k.wt; k.wawt ]) in
    let h = Mi355x.pinocchio_upload ~n ~m l r o ~mid ~g1 ~g2 in
    if !derive_lagrange_on_upload then Mi355x.(check (zk_pinocchio_pk_derive_lagrange h));
    Gc.finalise (fun _ -> ignore (Mi355x.zk_pinocchio_pk_free h)) k;
    Handles.replace handles k h;
    h

  let handle_of (qap : qap) (k : pkey) =
    match Handles.find_opt handles k with
    | Some h -> h
    | None ->
        let vars = keys qap.QAP.v in
        let n = Poly.degree qap.QAP.target in
        upload vars (matrices_of_qap (index_of_vars vars) qap n) n k

  (* ---------------------------------------------------------------- keygen *)

  let keygen rng (circuit : circuit) (qap : qap) : pkey * vkey =
    let n = Poly.degree qap.QAP.target in
    let rv = Fr.gen rng in
    let rw = Fr.gen rng in
    let s = Fr.gen rng in
    let av = Fr.gen rng in
    let aw = Fr.gen rng in
    let ay = Fr.gen rng in
    let b = Fr.gen rng in
    let gm = Fr.gen rng in
    let ry = Fr.(rv * rw) in
    let t = Poly.apply qap.QAP.target s in
    let at_s polys = Var.Map.map (fun p -> Poly.apply p s) polys in
    let v_s = at_s qap.QAP.v and w_s = at_s qap.QAP.w and y_s = at_s qap.QAP.y in
    let mids = circuit.Circuit.mids and ios = Circuit.ios circuit in
    let scaled set f m = values (Var.Map.map (fun x -> Fr.(x * f)) (Var.Map.restrict set m)) in
    let s_powers =
      let rec go acc x i = if i > n then List.rev acc else go (x :: acc) Fr.(x * s) (i + 1) in
      go [] Fr.one 0
    in
    let vt = Fr.(rv * t) and wt = Fr.(rw * t) and yt = Fr.(ry * t) in
    let n_mid = Var.Set.cardinal mids and n_io = Var.Set.cardinal ios and m = Var.Map.cardinal v_s in
    let combined =
      List.map
        (fun k -> Fr.(((rv * Var.Map.find k v_s) + (rw * Var.Map.find k w_s) + (ry * Var.Map.find k y_s)) * b))
        (Var.Set.elements mids)
    in
    let e1 =
      scaled mids rv v_s @ scaled mids ry y_s
      @ scaled mids Fr.(rv * av) v_s
      @ scaled mids Fr.(ry * ay) y_s
      @ combined @ s_powers @ values v_s @ values w_s
      @ [ vt; yt; Fr.(vt * av); Fr.(yt * ay); Fr.(vt * b); Fr.(wt * b); Fr.(yt * b) ]
      (* verification key, G1 part *)
      @ [ aw; Fr.(gm * b) ]
      @ scaled ios rv v_s @ scaled ios ry y_s
    in
    let e2 =
      scaled mids rw w_s
      @ scaled mids Fr.(rw * aw) w_s
      @ s_powers
      @ [ wt; Fr.(wt * aw) ]
      (* verification key, G2 part *)
      @ [ av; ay; gm; Fr.(gm * b); yt ]
      @ scaled ios rw w_s
    in
    let p1 = Mi355x.of_fr_many ~g2:false (fr_bytes e1) and p2 = Mi355x.of_fr_many ~g2:true (fr_bytes e2) in
    let map_of set at buf off = Var.Map.of_list (List.mapi (fun i k -> (k, at buf (off + i))) (Var.Set.elements set)) in
    let all_vars = Var.Map.domain v_s in
    let o_si = 5 * n_mid in
    let o_all = o_si + n + 1 in
    let o_single = o_all + (2 * m) in
    let o_vk = o_single + 7 in
    let o2_si = 2 * n_mid in
    let o2_single = o2_si + n + 1 in
    let o2_vk = o2_single + 2 in
    let pkey : pkey =
      { vv = map_of mids g1_at p1 0;
        yy = map_of mids g1_at p1 n_mid;
        vav = map_of mids g1_at p1 (2 * n_mid);
        yay = map_of mids g1_at p1 (3 * n_mid);
        bvwy = map_of mids g1_at p1 (4 * n_mid);
        si = List.init (n + 1) (fun i -> g1_at p1 (o_si + i));
        v_all = map_of all_vars g1_at p1 o_all;
        w_all = map_of all_vars g1_at p1 (o_all + m);
        vt = g1_at p1 o_single;
        yt = g1_at p1 (o_single + 1);
        vavt = g1_at p1 (o_single + 2);
        yayt = g1_at p1 (o_single + 3);
        vbt = g1_at p1 (o_single + 4);
        wbt = g1_at p1 (o_single + 5);
        ybt = g1_at p1 (o_single + 6);
        ww = map_of mids g2_at p2 0;
        waw = map_of mids g2_at p2 n_mid;
        si2 = List.init (n + 1) (fun i -> g2_at p2 (o2_si + i));
        wt = g2_at p2 o2_single;
        wawt = g2_at p2 (o2_single + 1)
      }
    in
    let vkey : vkey =
      { one = G1.one;
        one2 = G2.one;
        aw = g1_at p1 o_vk;
        bgm = g1_at p1 (o_vk + 1);
        vv_io = map_of ios g1_at p1 (o_vk + 2);
        yy_io = map_of ios g1_at p1 (o_vk + 2 + n_io);
        av = g2_at p2 o2_vk;
        ay = g2_at p2 (o2_vk + 1);
        gm2 = g2_at p2 (o2_vk + 2);
        bgm2 = g2_at p2 (o2_vk + 3);
        yt = g2_at p2 (o2_vk + 4);
        ww_io = map_of ios g2_at p2 (o2_vk + 5)
      }
    in
    let vars = keys qap.QAP.v in
    ignore (upload vars (matrices_of_gates (index_of_vars vars) circuit.Circuit.gates) n pkey);
    (pkey, vkey)

  (* ---------------------------------------------------------------- prove *)

  let prove_with (qap : qap) (k : pkey) (sol : Fr.t Var.Map.t) dv dw dy : proof =
    let handle = handle_of qap k in
    if not (Var.Set.equal (Var.Map.domain sol) (Var.Map.domain qap.QAP.v)) then assert false;
    let out = Bytes.create 960 in
    Mi355x.(
      check
        (zk_pinocchio_prove handle
           (bytes_start (fr_bytes (values sol)))
           (bytes_start (Fr.to_bytes dv))
           (bytes_start (Fr.to_bytes dw))
           (bytes_start (Fr.to_bytes dy))
           (bytes_start out)));
    (* Compute.proof, pinocchio.ml:195-208: vv | ww (G2) | yy | h | vavv | waww (G2) | yayy | bvwy *)
    let g1 off = G1.of_bytes_exn (Bytes.sub out off 96) and g2 off = G2.of_bytes_exn (Bytes.sub out off 192) in
    { vv = g1 0; ww = g2 96; yy = g1 288; h = g1 384; vavv = g1 480; waww = g2 576; yayy = g1 768; bvwy = g1 864 }

  let verify_with (ios : Fr.t Var.Map.t) (vk : vkey) (p : proof) : bool =
    (* the reference asserts equal domains before each of its three sums (pinocchio.ml:372,381,390) *)
    assert (Var.Set.equal (Var.Map.domain ios) (Var.Map.domain vk.vv_io));
    assert (Var.Set.equal (Var.Map.domain ios) (Var.Map.domain vk.ww_io));
    assert (Var.Set.equal (Var.Map.domain ios) (Var.Map.domain vk.yy_io));
    let vk_g1 = g1_bytes ((vk.one :: vk.aw :: vk.bgm :: values vk.vv_io) @ values vk.yy_io) in
    let vk_g2 = g2_bytes ((vk.one2 :: vk.av :: vk.ay :: vk.gm2 :: vk.bgm2 :: vk.yt :: values vk.ww_io)) in
    let proof_bytes =
      Mi355x.cat
        [ G1.to_bytes p.vv; G2.to_bytes p.ww; G1.to_bytes p.yy; G1.to_bytes p.h; G1.to_bytes p.vavv; G2.to_bytes p.waww;
          G1.to_bytes p.yayy; G1.to_bytes p.bvwy ]
    in
    let ok = Ctypes.allocate Ctypes.int 0 in
    Mi355x.(
      check
        (zk_pinocchio_verify (bytes_start vk_g1) (bytes_start vk_g2)
           (bytes_start (fr_bytes (values ios)))
           (sz (Var.Map.cardinal ios))
           (bytes_start proof_bytes) ok));
    Ctypes.( !@ ) ok <> 0

  module NonZK = struct
    type f = C.Fr.t
    type nonrec circuit = circuit
    type nonrec qap = qap
    type nonrec pkey = pkey [@@deriving yojson]
    type nonrec vkey = vkey [@@deriving yojson]
    type nonrec proof = proof [@@deriving yojson]

    let keygen = keygen
    let prove _rng qap pkey sol = prove_with qap pkey sol Fr.zero Fr.zero Fr.zero
    let verify input_output vkey proof = verify_with input_output vkey proof
  end

  module ZK = struct
    type f = C.Fr.t
    type nonrec circuit = circuit
    type nonrec qap = qap
    type nonrec pkey = pkey [@@deriving yojson]
    type nonrec vkey = vkey [@@deriving yojson]
    type nonrec proof = proof [@@deriving yojson]

    let keygen = keygen

    let prove rng qap pkey sol =
      let dv = Fr.gen rng in
      let dw = Fr.gen rng in
      let dy = Fr.gen rng in
      prove_with qap pkey sol dv dw dy

    let verify = NonZK.verify
  end
end
