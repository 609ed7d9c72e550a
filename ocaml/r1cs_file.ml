(* r1cs_file.ml -- writers of the `.r1cs` / `.wit` interchange files (zukelang_amd/r1cs_file.py defines and reads them).

   The reference has no circuit file format: a circuit is an OCaml value (src/lib/zk/circuit.ml:73-75).  These two files hold
   exactly what zk_groth16_pk_upload / zk_groth16_prove take, for hosts that do not link the library or to hand a circuit to
   another machine.  Little-endian, sections 8-byte aligned (a C host can mmap the file and point a zk_csr into it).
   Gates in Gate.Set.elements order (= the ids of QAP.ml:22), variables in Var.compare order.  Goes to src/lib/zk/. *)

module Make (F : sig
  include Field.COMPARABLE

  val to_bytes : t -> bytes
end) =
struct
  module Circuit = Circuit.Make (F)

  let write_r1cs oc (circuit : Circuit.t) =
    let gates = Circuit.Gate.Set.elements circuit.Circuit.gates in
    let vars = Var.Set.elements (Circuit.vars circuit.Circuit.gates) in
    let index = Var.Map.of_list (List.mapi (fun i v -> (v, i)) vars) in
    let b = Buffer.create 4096 in
    let u32 x = Buffer.add_int32_le b (Int32.of_int x) and u64 x = Buffer.add_int64_le b (Int64.of_int x) in
    let pad () =
      while Buffer.length b mod 8 <> 0 do
        Buffer.add_char b '\000'
      done
    in
    Buffer.add_string b "ZKR1CS\000\001";
    u32 1;
    u32 32;
    let rows sel = List.map (fun g -> Var.Map.bindings (sel g)) gates in
    let l = rows (fun (g : Circuit.Gate.t) -> g.l)
    and r = rows (fun (g : Circuit.Gate.t) -> g.r)
    and o = rows (fun (g : Circuit.Gate.t) -> g.lhs) in
    let nnz m = List.fold_left (fun a row -> a + List.length row) 0 m in
    u64 (List.length gates);
    u64 (List.length vars);
    u64 (nnz l);
    u64 (nnz r);
    u64 (nnz o);
    List.iter
      (fun (v : Var.t) ->
        let name, id = (v :> string * int) in
        u32 id;
        u32 (String.length name);
        Buffer.add_string b name)
      vars;
    pad ();
    List.iter (fun v -> Buffer.add_char b (if Var.Set.mem v circuit.Circuit.mids then '\001' else '\000')) vars;
    pad ();
    List.iter
      (fun m ->
        let acc = ref 0 in
        u32 0;
        List.iter
          (fun row ->
            acc := !acc + List.length row;
            u32 !acc)
          m;
        pad ();
        List.iter (List.iter (fun (v, _) -> u32 (Var.Map.find v index))) m;
        pad ();
        List.iter (List.iter (fun (_, c) -> Buffer.add_bytes b (F.to_bytes c))) m)
      [ l; r; o ];
    Buffer.output_buffer oc b

  let write_witness oc (sol : F.t Var.Map.t) =
    let b = Buffer.create 4096 in
    Buffer.add_string b "ZKWIT\000\000\001";
    Buffer.add_int32_le b 1l;
    Buffer.add_int32_le b 32l;
    Buffer.add_int64_le b (Int64.of_int (Var.Map.cardinal sol));
    Var.Map.iter (fun _ v -> Buffer.add_bytes b (F.to_bytes v)) sol;
    Buffer.output_buffer oc b
end
