//! Bit-exact parity of the MI355X path with the real `ecfft` crate — the parity suite this repo could not run itself
//! (no Rust in the build image).  UNCOMPILED SOURCE.  Shapes follow the reference's own tests (src/lib.rs:108-186, 239-278)
//! and bench (benches/fftree.rs:19-62); every comparison is `assert_eq!` on `Vec<F>`, i.e. on canonical field elements.
use ark_ff::{One, UniformRand};
use ecfft::{FFTree, FftreeField, Moiety};
use ecfft_hip::{ffi, HipFFTree, HipField};
use rand::rngs::StdRng;
use rand::SeedableRng;

fn rand_vec<F: HipField>(n: usize, seed: u8) -> Vec<F> {
    let mut rng = StdRng::from_seed([seed; 32]);
    (0..n).map(|_| F::rand(&mut rng)).collect()
}

/// the one encoding fact the C ABI relies on and this repo could not confirm against ark-ff itself:
/// secp256k1 `Fp::one()` in memory = 2^256 mod p = 0x1000003d1 as little-endian u64 limbs; m31 `Fp(1)` = 1u32
#[test]
fn in_memory_encoding_is_what_the_abi_assumes() {
    let one = ecfft::secp256k1::Fp::one();
    let bytes: [u8; 32] = unsafe { core::mem::transmute(one) };
    let mut want = [0u8; 32];
    want[..5].copy_from_slice(&[0xd1, 0x03, 0x00, 0x00, 0x01]);
    assert_eq!(bytes, want, "ark-ff MontBackend limb encoding differs from x*2^256 mod p, little-endian");
    let m = ecfft::m31::Fp::one();
    let b4: [u8; 4] = unsafe { core::mem::transmute(m) };
    assert_eq!(b4, [1, 0, 0, 0]);
}

fn all_algorithms_match<F: HipField>(log_tree: u32) {
    let n_tree = 1usize << log_tree;
    let cpu: FFTree<F> = F::build_fftree(n_tree).unwrap();
    let gpu: HipFFTree<F> = HipFFTree::build_fftree(n_tree).unwrap();
    // the two constructions agree on every pub table of every subtree (src/fftree.rs:24-38)
    let mut sub: Option<&FFTree<F>> = Some(&cpu);
    while let Some(t) = sub {
        let m = t.f.leaves().len();
        assert_eq!(gpu.eval_domain(m), t.f.leaves());
        assert_eq!(gpu.table(m, ffi::TBL_XNN_S), t.xnn_s);
        assert_eq!(gpu.table(m, ffi::TBL_XNN_S_INV), t.xnn_s_inv);
        assert_eq!(gpu.table(m, ffi::TBL_Z0_S1), t.z0_s1);
        assert_eq!(gpu.table(m, ffi::TBL_Z1_S0), t.z1_s0);
        assert_eq!(gpu.table(m, ffi::TBL_Z0_INV_S1), t.z0_inv_s1);
        assert_eq!(gpu.table(m, ffi::TBL_Z1_INV_S0), t.z1_inv_s0);
        assert_eq!(gpu.table(m, ffi::TBL_Z0Z0_REM_XNN_S), t.z0z0_rem_xnn_s);
        assert_eq!(gpu.table(m, ffi::TBL_Z1Z1_REM_XNN_S), t.z1z1_rem_xnn_s);
        sub = t.subtree.as_deref();
    }
    let mut n = 1usize;
    while n <= n_tree {
        let v: Vec<F> = rand_vec(n, 1 + n.trailing_zeros() as u8);
        assert_eq!(gpu.enter(&v), cpu.enter(&v), "ENTER n={n}");
        assert_eq!(gpu.exit(&v), cpu.exit(&v), "EXIT n={n}");
        assert_eq!(gpu.exit(&gpu.enter(&v)), v, "round trip n={n}");
        assert_eq!(gpu.degree(&v), cpu.degree(&v), "DEGREE n={n}");
        if 2 * n <= n_tree {
            for m in [Moiety::S0, Moiety::S1] {
                assert_eq!(gpu.extend(&v, m), cpu.extend(&v, m), "EXTEND n={n}");
                assert_eq!(gpu.mextend(&v, m), cpu.mextend(&v, m), "MEXTEND n={n}");
            }
            assert_eq!(gpu.vanish(&v), cpu.vanish(&v), "VANISH n={n}");
        }
        if n >= 2 {
            let t = cpu.subtree_with_size(n);
            assert_eq!(gpu.redc_z0(&v, &t.xnn_s), cpu.redc_z0(&v, &t.xnn_s), "REDC_z0 n={n}");
            assert_eq!(gpu.redc_z1(&v, &t.xnn_s), cpu.redc_z1(&v, &t.xnn_s), "REDC_z1 n={n}");
            assert_eq!(gpu.modular_reduce(&v, &t.xnn_s, &t.z0z0_rem_xnn_s), cpu.modular_reduce(&v, &t.xnn_s, &t.z0z0_rem_xnn_s), "MOD n={n}");
        }
        n *= 2;
    }
    // a device tree mirrored from the CPU tree's point set (FFTree::new path) behaves identically
    let mirrored = HipFFTree::from_cpu_tree(&cpu, 0);
    let v: Vec<F> = rand_vec(n_tree, 99);
    assert_eq!(mirrored.enter(&v), cpu.enter(&v));
}

#[test]
fn secp256k1_matches_the_crate() {
    all_algorithms_match::<ecfft::secp256k1::Fp>(12);
}

#[test]
fn m31_matches_the_crate() {
    all_algorithms_match::<ecfft::m31::Fp>(14);
}

#[test]
#[should_panic(expected = "FFTree is too small")]
fn too_small_tree_panics_like_the_reference() {
    let gpu: HipFFTree<ecfft::m31::Fp> = HipFFTree::build_fftree(64).unwrap();
    let v: Vec<ecfft::m31::Fp> = rand_vec(128, 1);
    let _ = gpu.enter(&v);
}

#[test]
fn build_fftree_returns_none_beyond_two_adicity() {
    assert!(HipFFTree::<ecfft::m31::Fp>::build_fftree(1 << 29).is_none()); // src/ec.rs:513-515
}

/// The wire format: bytes written by the crate load into the device tree, and the device tree writes the crate's bytes back
/// (src/fftree.rs:507-660; the reference's own tests: deserialized_{un,}compressed_tree_works, src/lib.rs:154-186).
#[test]
fn wire_format_round_trips_with_the_crate() {
    use ark_serialize::{CanonicalDeserialize, CanonicalSerialize, Compress};
    type F = ecfft::m31::Fp;
    let cpu = F::build_fftree(64).unwrap();
    for compress in [Compress::Yes, Compress::No] {
        let mut bytes = Vec::new();
        cpu.serialize_with_mode(&mut bytes, compress).unwrap();
        let gpu = HipFFTree::<F>::deserialize(&bytes, compress, 0, true).expect("crate-written file must load and verify");
        assert_eq!(gpu.serialize(compress), bytes, "device tree re-serialises to the crate's bytes");
        let back = ecfft::FFTree::<F>::deserialize_with_mode(&gpu.serialize(compress)[..], compress, ark_serialize::Validate::Yes).unwrap();
        let v: Vec<F> = rand_vec(64, 5);
        assert_eq!(back.enter(&v), cpu.enter(&v));
        assert_eq!(gpu.enter(&v), cpu.enter(&v));
    }
    let mut bytes = Vec::new();
    cpu.serialize_compressed(&mut bytes).unwrap();
    assert!(HipFFTree::<F>::deserialize(&bytes[..bytes.len() - 1], Compress::Yes, 0, true).is_none(), "truncated file");
}
