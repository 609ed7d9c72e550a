(* mi355x.ml -- ctypes binding of libzkmi355x (include/zkmi355x.h), the MI355X prove path.

   Goes to src/lib/zk/ of the zukelang tree (library `zk`, plus `ctypes ctypes.foreign` in its dune stanza: see ocaml/dune).
   Every entry point an OCaml host needs is bound here; the header says which reference interface each one replaces.
   This file cannot be compiled in the image it was written in (no OCaml toolchain there); tests/test_ocaml_binding.py
   holds every symbol name and every arity below against the header and the built library.

   Conventions of the C side: Fr = 32 B little-endian (Bls12_381.Fr.to_bytes), G1 / G2 = 96 / 192 B uncompressed
   (G1.to_bytes / G2.to_bytes), all buffers caller-owned, 0 = ok, negative = error code. *)

open Ctypes
open Foreign

let lib =
  let file = try Sys.getenv "ZK_LIBZKMI355X_PATH" with Not_found -> "libzkmi355x.so" in
  Dl.dlopen ~filename:file ~flags:[ Dl.RTLD_NOW ]

let fn name typ = foreign ~from:lib name typ

(* ------------------------------------------------------------------ small helpers *)

let u32 = Unsigned.UInt32.of_int
let u64 = Unsigned.UInt64.of_int
let sz = Unsigned.Size_t.of_int
let bytes_start = ocaml_bytes_start
let cat = Bytes.concat Bytes.empty
let null_bytes : char ptr = from_voidp char null

(* an OCaml buffer copied into C memory, for the few arguments that live in a C struct *)
let carray_of_bytes (b : bytes) : char CArray.t =
  let a = CArray.make char (max 1 (Bytes.length b)) in
  Bytes.iteri (fun i c -> CArray.set a i c) b;
  a

(* ------------------------------------------------------------------ errors
   The reference raises exceptions; the C side returns codes (header, "Errors").
     ZK_ERR_APPLY_POWERS (-6)  -> Invalid_argument "apply_powers"        curve.ml:116
     ZK_ERR_REMAINDER (-4)     -> Assert_failure (QAP.ml:134: assert (Polynomial.is_zero rem))
     ZK_ERR_DOMAIN (-8)        -> Assert_failure (curve.ml:96-100: assert false in G.dot)
     ZK_ERR_NOT_ON_CURVE (-2)  -> Bls12_381.G1.Not_on_curve              what of_bytes_exn raises on such bytes (curve.ml:199-212)
     ZK_ERR_SCALAR_RANGE (-3)  -> Bls12_381.Fr.Not_in_field              what Fr.of_bytes_exn raises on a value >= r
     anything else             -> Failure with the library's text
   Not_on_curve / Not_in_field are the exception constructors of opam bls12-381 6.1.0 (its source is not vendored in zukelang;
   if a later version renames them this is the one place to touch). *)

let zk_strerror = fn "zk_strerror" (int @-> returning string)
let zk_last_error = fn "zk_last_error" (void @-> returning string)

let check rc =
  if rc = 0 then ()
  else
    let detail () = zk_strerror rc ^ ": " ^ zk_last_error () in
    match rc with
    | -6 -> invalid_arg "apply_powers"
    | -4 -> raise (Assert_failure ("QAP.ml", 134, 4))
    | -8 -> raise (Assert_failure ("curve.ml", 100, 8))
    | -2 -> raise (Bls12_381.G1.Not_on_curve (Bytes.of_string (detail ())))
    | -3 -> raise (Bls12_381.Fr.Not_in_field (Bytes.of_string (detail ())))
    | _ -> failwith (detail ())

(* ------------------------------------------------------------------ devices and options *)

let zk_device_count = fn "zk_device_count" (void @-> returning int)
let zk_init = fn "zk_init" (int @-> returning int)
let zk_shutdown = fn "zk_shutdown" (void @-> returning int)
let zk_set_devices = fn "zk_set_devices" (uint64_t @-> returning int)
let zk_sync = fn "zk_sync" (void @-> returning int)

(* knobs that are not test-only, by name (the environment stays a fallback): see the header *)
let zk_set_option = fn "zk_set_option" (string @-> string @-> returning int)

(* all MI355X of the node behind every key uploaded afterwards (header, "multi-device keys") *)
let use_all_devices () =
  let n = zk_device_count () in
  if n > 1 then check (zk_set_devices (Unsigned.UInt64.of_int ((1 lsl n) - 1)))

(* ------------------------------------------------------------------ Fr stage: FFT.ml:69-105 *)

let zk_fr_ntt = fn "zk_fr_ntt" (ocaml_bytes @-> uint32_t @-> int @-> returning int)

let zk_fr_poly_mul =
  fn "zk_fr_poly_mul" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> size_t @-> ocaml_bytes @-> ptr size_t @-> returning int)

(* ------------------------------------------------------------------ curve plugin seam: curve.ml:94-118,180 *)

let zk_msm_g1 =
  fn "zk_msm_g1" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> size_t @-> uint32_t @-> ocaml_bytes @-> returning int)

let zk_msm_g2 =
  fn "zk_msm_g2" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> size_t @-> uint32_t @-> ocaml_bytes @-> returning int)

let zk_g1_of_fr = fn "zk_g1_of_fr" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> returning int)
let zk_g2_of_fr = fn "zk_g2_of_fr" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> returning int)
let zk_g1_powers = fn "zk_g1_powers" (uint32_t @-> ocaml_bytes @-> ocaml_bytes @-> returning int)
let zk_g2_powers = fn "zk_g2_powers" (uint32_t @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

(* sum_i scalars_i * points_i over byte strings; 96 or 192 bytes per point *)
let msm ~g2 (points : bytes) (scalars : bytes) : bytes =
  let psize = if g2 then 192 else 96 in
  let out = Bytes.create psize in
  let f = if g2 then zk_msm_g2 else zk_msm_g1 in
  check
    (f (bytes_start points) (sz (Bytes.length points / psize)) (bytes_start scalars) (sz (Bytes.length scalars / 32)) (u32 0)
       (bytes_start out));
  out

(* [g * s_0; g * s_1; ...] for the group's generator g: G.of_Fr mapped over a list (curve.ml:180), one kernel launch *)
let of_fr_many ~g2 (scalars : bytes) : bytes =
  let psize = if g2 then 192 else 96 in
  let n = Bytes.length scalars / 32 in
  let out = Bytes.create (max 1 (psize * n)) in
  if n > 0 then check ((if g2 then zk_g2_of_fr else zk_g1_of_fr) (bytes_start scalars) (sz n) (bytes_start out));
  Bytes.sub out 0 (psize * n)

(* of_compressed_bytes_exn over a whole list on the GPU (a key read from the reference's JSON: every point is compressed there): the compressed points
   back to back in, the uncompressed points back to back out; Not_on_curve / Invalid_argument as the one-point functions of Bls12_381 raise *)
let zk_g1_decompress_batch = fn "zk_g1_decompress_batch" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> returning int)
let zk_g2_decompress_batch = fn "zk_g2_decompress_batch" (ocaml_bytes @-> size_t @-> ocaml_bytes @-> returning int)

let decompress_many ~g2 (comp : bytes) : bytes =
  let csize = if g2 then 96 else 48 in
  let n = Bytes.length comp / csize in
  if Bytes.length comp <> n * csize then invalid_arg "decompress_many";
  let out = Bytes.create (max 1 (2 * csize * n)) in
  if n > 0 then check ((if g2 then zk_g2_decompress_batch else zk_g1_decompress_batch) (bytes_start comp) (sz n) (bytes_start out));
  Bytes.sub out 0 (2 * csize * n)

let powers ~g2 d (s : bytes) : bytes =
  let psize = if g2 then 192 else 96 in
  let out = Bytes.create (psize * (d + 1)) in
  check ((if g2 then zk_g2_powers else zk_g1_powers) (u32 d) (bytes_start s) (bytes_start out));
  out

(* ------------------------------------------------------------------ circuits: three CSR matrices (header, zk_csr) *)

type csr
let csr : csr structure typ = structure "zk_csr"
let csr_row_ptr = field csr "row_ptr" (ptr uint32_t)
let csr_col = field csr "col" (ptr uint32_t)
let csr_val = field csr "val" (ptr char)
let () = seal csr

(* One matrix, built from rows of (column, 32-byte coefficient) in ascending column order; the record keeps the C arrays alive. *)
type matrix = { c : csr structure; keep_ptr : Unsigned.uint32 CArray.t; keep_col : Unsigned.uint32 CArray.t; keep_val : char CArray.t }

let matrix_of_rows (rows : (int * bytes) list list) : matrix =
  let nnz = List.fold_left (fun a r -> a + List.length r) 0 rows in
  let keep_ptr = CArray.make uint32_t (List.length rows + 1) in
  let keep_col = CArray.make uint32_t (max 1 nnz) in
  let keep_val = CArray.make char (max 1 (32 * nnz)) in
  CArray.set keep_ptr 0 (u32 0);
  let e = ref 0 in
  List.iteri
    (fun g row ->
      List.iter
        (fun (k, coeff) ->
          CArray.set keep_col !e (u32 k);
          Bytes.iteri (fun i ch -> CArray.set keep_val ((32 * !e) + i) ch) coeff;
          incr e)
        row;
      CArray.set keep_ptr (g + 1) (u32 !e))
    rows;
  let c = make csr in
  setf c csr_row_ptr (CArray.start keep_ptr);
  setf c csr_col (CArray.start keep_col);
  setf c csr_val (CArray.start keep_val);
  { c; keep_ptr; keep_col; keep_val }

(* y = M x over Fr for one sparse matrix (header: zk_fr_spmv): QAP.eval's sums at the gates' points, or -- M transposed, x = the Lagrange basis at
   tau -- every u_k(tau) of a keygen at once, where the reference runs Poly.apply per variable *)
let zk_fr_spmv = fn "zk_fr_spmv" (uint32_t @-> uint32_t @-> ptr csr @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

(* ------------------------------------------------------------------ Groth16: groth16.ml:24-34,116-161,235-237 *)

let zk_groth16_pk_upload =
  fn "zk_groth16_pk_upload"
    (uint32_t @-> uint32_t @-> ptr csr @-> ptr csr @-> ptr csr @-> ocaml_bytes @-> ocaml_bytes @-> size_t @-> ocaml_bytes @-> size_t
   @-> ptr uint64_t @-> returning int)

let zk_groth16_pk_derive_lagrange = fn "zk_groth16_pk_derive_lagrange" (uint64_t @-> returning int)
let zk_groth16_pk_free = fn "zk_groth16_pk_free" (uint64_t @-> returning int)

let zk_groth16_prove =
  fn "zk_groth16_prove" (uint64_t @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

(* the same entry point taking the witness made resident by zk_groth16_set_witness (sol = NULL) *)
let zk_groth16_prove_resident =
  fn "zk_groth16_prove" (uint64_t @-> ptr char @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

let zk_groth16_reserve_slots = fn "zk_groth16_reserve_slots" (uint64_t @-> uint32_t @-> returning int)
let zk_groth16_set_witness = fn "zk_groth16_set_witness" (uint64_t @-> ocaml_bytes @-> returning int)

let zk_groth16_prove_async =
  fn "zk_groth16_prove_async" (uint64_t @-> ptr char @-> ocaml_bytes @-> ocaml_bytes @-> uint32_t @-> returning int)

let zk_groth16_prove_wait = fn "zk_groth16_prove_wait" (uint64_t @-> uint32_t @-> ocaml_bytes @-> returning int)

let zk_groth16_qap_eval =
  fn "zk_groth16_qap_eval" (uint64_t @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

let zk_groth16_verify =
  fn "zk_groth16_verify"
    (ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> size_t @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> ptr int @-> returning int)

let groth16_upload ~n ~m (l : matrix) (r : matrix) (o : matrix) ~(mid : bytes) ~(g1 : bytes) ~(g2 : bytes) : Unsigned.UInt64.t =
  let h = allocate uint64_t Unsigned.UInt64.zero in
  check
    (zk_groth16_pk_upload (u32 n) (u32 m) (addr l.c) (addr r.c) (addr o.c) (bytes_start mid) (bytes_start g1)
       (sz (Bytes.length g1 / 96))
       (bytes_start g2)
       (sz (Bytes.length g2 / 192))
       h);
  !@h

(* ------------------------------------------------------------------ Pinocchio Protocol 2: pinocchio.ml:37-60,210-248,427-514 *)

let zk_pinocchio_pk_upload =
  fn "zk_pinocchio_pk_upload"
    (uint32_t @-> uint32_t @-> ptr csr @-> ptr csr @-> ptr csr @-> ocaml_bytes @-> ocaml_bytes @-> size_t @-> ocaml_bytes @-> size_t
   @-> ptr uint64_t @-> returning int)

let zk_pinocchio_pk_derive_lagrange = fn "zk_pinocchio_pk_derive_lagrange" (uint64_t @-> returning int)
let zk_pinocchio_pk_free = fn "zk_pinocchio_pk_free" (uint64_t @-> returning int)

let zk_pinocchio_prove =
  fn "zk_pinocchio_prove"
    (uint64_t @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> returning int)

let zk_pinocchio_reserve_slots = fn "zk_pinocchio_reserve_slots" (uint64_t @-> uint32_t @-> returning int)
let zk_pinocchio_set_witness = fn "zk_pinocchio_set_witness" (uint64_t @-> ocaml_bytes @-> returning int)

let zk_pinocchio_prove_async =
  fn "zk_pinocchio_prove_async"
    (uint64_t @-> ptr char @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> uint32_t @-> returning int)

let zk_pinocchio_prove_wait = fn "zk_pinocchio_prove_wait" (uint64_t @-> uint32_t @-> ocaml_bytes @-> returning int)

let zk_pinocchio_verify =
  fn "zk_pinocchio_verify" (ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> size_t @-> ocaml_bytes @-> ptr int @-> returning int)

let pinocchio_upload ~n ~m (l : matrix) (r : matrix) (o : matrix) ~(mid : bytes) ~(g1 : bytes) ~(g2 : bytes) : Unsigned.UInt64.t =
  let h = allocate uint64_t Unsigned.UInt64.zero in
  check
    (zk_pinocchio_pk_upload (u32 n) (u32 m) (addr l.c) (addr r.c) (addr o.c) (bytes_start mid) (bytes_start g1)
       (sz (Bytes.length g1 / 96))
       (bytes_start g2)
       (sz (Bytes.length g2 / 192))
       h);
  !@h

(* ------------------------------------------------------------------ N GPUs, one process per GPU (header, "point-sharded multi-GPU prove") *)

let zk_groth16_pk_shard = fn "zk_groth16_pk_shard" (uint64_t @-> uint32_t @-> uint32_t @-> returning int)

let zk_groth16_prove_partial_async =
  fn "zk_groth16_prove_partial_async" (uint64_t @-> ocaml_bytes @-> ocaml_bytes @-> ocaml_bytes @-> uint32_t @-> returning int)

let zk_groth16_prove_partial_wait_device =
  fn "zk_groth16_prove_partial_wait_device" (uint64_t @-> uint32_t @-> ptr void @-> returning int)

let zk_groth16_combine_device =
  fn "zk_groth16_combine_device" (ptr void @-> size_t @-> uint32_t @-> ocaml_bytes @-> returning int)

let zk_device_malloc = fn "zk_device_malloc" (size_t @-> ptr (ptr void) @-> returning int)
let zk_device_free = fn "zk_device_free" (ptr void @-> returning int)
