//! Dumps, from the REAL ecfft crate, everything this repo had to restate without being able to observe it.
//! Output: `key = hex` lines (see README.md).  UNCOMPILED SOURCE.
use ark_ff::PrimeField;
use ark_serialize::CanonicalSerialize;
use ecfft::{FFTree, FftreeField, Moiety};

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{b:02x}")).collect()
}

/// raw in-memory bytes of a slice of field elements (what crosses the C ABI)
fn mem<F: Copy>(v: &[F]) -> String {
    let p = v.as_ptr() as *const u8;
    hex(unsafe { core::slice::from_raw_parts(p, core::mem::size_of_val(v)) })
}

fn dump<F: FftreeField>(name: &str) {
    // 1. element encoding: small integers and -1
    let small: Vec<F> = vec![F::from(0u64), F::from(1u64), F::from(2u64), F::from(977u64), -F::from(1u64)];
    println!("{name}.size_of = {:02x}", core::mem::size_of::<F>());
    println!("{name}.mem.0_1_2_977_minus1 = {}", mem(&small));
    // 2. ENTER / EXIT / EXTEND known answers on [1, 2, .., n]
    for n in [4usize, 64] {
        let tree: FFTree<F> = F::build_fftree(n).unwrap();
        let coeffs: Vec<F> = (1..=n as u64).map(F::from).collect();
        let evals = tree.enter(&coeffs);
        println!("{name}.n{n}.leaves.mem = {}", mem(tree.f.leaves()));
        println!("{name}.n{n}.enter_1_to_n.mem = {}", mem(&evals));
        println!("{name}.n{n}.exit_1_to_n.mem = {}", mem(&tree.exit(&coeffs)));
        if n >= 4 {
            let half: Vec<F> = coeffs[..n / 2].to_vec();
            println!("{name}.n{n}.extend_s1_1_to_half.mem = {}", mem(&tree.extend(&half, Moiety::S1)));
            println!("{name}.n{n}.extend_s0_1_to_half.mem = {}", mem(&tree.extend(&half, Moiety::S0)));
        }
        println!("{name}.n{n}.xnn_s.mem = {}", mem(&tree.xnn_s));
        println!("{name}.n{n}.z0z0_rem_xnn_s.mem = {}", mem(&tree.z0z0_rem_xnn_s));
        // 3. wire format (src/fftree.rs:507-660)
        let mut c = Vec::new();
        tree.serialize_compressed(&mut c).unwrap();
        println!("{name}.n{n}.serialize_compressed = {}", hex(&c));
        let mut u = Vec::new();
        tree.serialize_uncompressed(&mut u).unwrap();
        println!("{name}.n{n}.serialize_uncompressed = {}", hex(&u));
        println!("{name}.n{n}.serialized_size_compressed = {:x}", tree.compressed_size());
    }
    println!("{name}.modulus_bits = {:x}", F::MODULUS_BIT_SIZE);
}

fn main() {
    dump::<ecfft::secp256k1::Fp>("secp256k1");
    dump::<ecfft::m31::Fp>("m31");
}
