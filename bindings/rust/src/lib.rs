//! `HipFFTree<F>`: the reference's `FFTree<F>` surface (src/fftree.rs:23-38, 123, 138, 164, 195, 227, 264-289, 313) backed by
//! the MI355X kernels through the C ABI of `include/ecfft_hip.h`.
//!
//! UNCOMPILED SOURCE — see README.md.  Element layout contract (ecfft_hip.h): `secp256k1::Fp` = `Fp256<MontBackend<_, 4>>`
//! = `[u64; 4]` Montgomery limbs, `m31::Fp` = one `u32`; both are passed as raw slices with no conversion, which is why
//! `HipField` is `unsafe` to implement.
use core::ffi::c_void;
use core::marker::PhantomData;

pub use ecfft::Moiety;

/// raw C ABI (include/ecfft_hip.h)
pub mod ffi {
    use core::ffi::c_void;
    #[repr(C)]
    pub struct EcfftCtx {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct EcfftComm {
        _p: [u8; 0],
    }
    pub const OK: i32 = 0;
    pub const ERR_NOT_POW2: i32 = 1;
    pub const ERR_TREE_TOO_SMALL: i32 = 2;
    pub const ERR_TREE_TOO_LARGE: i32 = 3;
    pub const ERR_HIP: i32 = 4;
    pub const ERR_BAD_ARG: i32 = 5;
    pub const MEM_HOST: i32 = 0;
    pub const MEM_DEVICE: i32 = 1;
    // ECFFT_TBL_*
    pub const TBL_F: i32 = 0;
    pub const TBL_RECOMBINE: i32 = 1;
    pub const TBL_DECOMPOSE: i32 = 2;
    pub const TBL_XNN_S: i32 = 3;
    pub const TBL_XNN_S_INV: i32 = 4;
    pub const TBL_Z0_S1: i32 = 5;
    pub const TBL_Z1_S0: i32 = 6;
    pub const TBL_Z0_INV_S1: i32 = 7;
    pub const TBL_Z1_INV_S0: i32 = 8;
    pub const TBL_Z0Z0_REM_XNN_S: i32 = 9;
    pub const TBL_Z1Z1_REM_XNN_S: i32 = 10;
    extern "C" {
        pub fn ecfft_elem_size(field: i32) -> usize;
        pub fn ecfft_build_fftree(field: i32, n: usize, device: i32, out: *mut *mut EcfftCtx) -> i32;
        pub fn ecfft_fftree_new(field: i32, leaves: *const c_void, n: usize, map_num3: *const c_void, map_den3: *const c_void, device: i32, out: *mut *mut EcfftCtx) -> i32;
        pub fn ecfft_ctx_destroy(ctx: *mut EcfftCtx);
        pub fn ecfft_tree_size(ctx: *const EcfftCtx) -> usize;
        pub fn ecfft_enter(ctx: *mut EcfftCtx, coeffs: *const c_void, evals: *mut c_void, n: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_exit(ctx: *mut EcfftCtx, evals: *const c_void, coeffs: *mut c_void, n: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_enter_many(ctx: *mut EcfftCtx, coeffs: *const c_void, evals: *mut c_void, n: usize, count: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_exit_many(ctx: *mut EcfftCtx, evals: *const c_void, coeffs: *mut c_void, n: usize, count: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_extend(ctx: *mut EcfftCtx, inp: *const c_void, out: *mut c_void, e: usize, moiety: i32, count: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_mextend(ctx: *mut EcfftCtx, inp: *const c_void, out: *mut c_void, e: usize, moiety: i32, count: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_redc(ctx: *mut EcfftCtx, evals: *const c_void, a: *const c_void, out: *mut c_void, n: usize, moiety: i32, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_modular_reduce(ctx: *mut EcfftCtx, evals: *const c_void, a: *const c_void, c: *const c_void, out: *mut c_void, n: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_vanish(ctx: *mut EcfftCtx, domain: *const c_void, out: *mut c_void, nd: usize, mem: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_degree(ctx: *mut EcfftCtx, evals: *const c_void, n: usize, mem: i32, stream: *mut c_void, degree: *mut usize) -> i32;
        pub fn ecfft_tree_table(ctx: *mut EcfftCtx, m: usize, which: i32, host_out: *mut c_void, cap: usize, count: *mut usize) -> i32;
        pub fn ecfft_ctx_device_bytes(ctx: *const EcfftCtx) -> usize;
        pub fn ecfft_ctx_trim(ctx: *mut EcfftCtx) -> i32;
        // FFTree wire format (impl CanonicalSerialize / CanonicalDeserialize for FFTree<F>, src/fftree.rs:507-660)
        pub fn ecfft_fftree_serialize(ctx: *mut EcfftCtx, compress: i32, buf: *mut c_void, cap: usize, len: *mut usize) -> i32;
        pub fn ecfft_fftree_deserialize(field: i32, bytes: *const c_void, len: usize, compress: i32, device: i32, verify: i32, out: *mut *mut EcfftCtx) -> i32;
        pub fn ecfft_tree_rational_maps(ctx: *mut EcfftCtx, map_num3_out: *mut c_void, map_den3_out: *mut c_void) -> i32;
        // one transform split over the GPUs of a node, one process per GPU (device pointers; see ecfft_hip.h)
        pub fn ecfft_comm_get_unique_id(id_out: *mut c_void) -> i32; // ECFFT_COMM_ID_BYTES = 128
        pub fn ecfft_comm_init_rank(id: *const c_void, world: i32, rank: i32, device: i32, out: *mut *mut EcfftComm) -> i32;
        pub fn ecfft_comm_destroy(comm: *mut EcfftComm);
        pub fn ecfft_comm_rank(comm: *const EcfftComm) -> i32;
        pub fn ecfft_comm_world(comm: *const EcfftComm) -> i32;
        pub fn ecfft_build_extend_shard(field: i32, e: usize, device: i32, world: i32, rank: i32, out: *mut *mut EcfftCtx) -> i32;
        pub fn ecfft_build_enter_shard(field: i32, n: usize, device: i32, world: i32, rank: i32, out: *mut *mut EcfftCtx) -> i32;
        pub fn ecfft_build_exit_shard(field: i32, n: usize, device: i32, comm: *mut EcfftComm, out: *mut *mut EcfftCtx) -> i32; // collective
        pub fn ecfft_build_exit_shard_opts(field: i32, n: usize, device: i32, comm: *mut EcfftComm, flags: i32, out: *mut *mut EcfftCtx) -> i32; // flags: 1 = ECFFT_EXIT_SHARD_MIN_MEMORY
        pub fn ecfft_comm_set_rccl_library(path: *const std::os::raw::c_char) -> i32; // before the first communicator; NULL = default
        pub fn ecfft_comm_abort(comm: *mut EcfftComm) -> i32;
        pub fn ecfft_comm_set_link_striping(comm: *mut EcfftComm, min_gain_bytes: usize) -> i32; // usize::MAX = never (the default since round 6), 0 = whenever it helps; same value on every rank (checked in the ranks' first vote); ERR_BAD_ARG once the communicator has carried an exchange
        pub fn ecfft_extend_sharded(ctx: *mut EcfftCtx, comm: *mut EcfftComm, input: *const c_void, out: *mut c_void, e: usize, moiety: i32, stream: *mut c_void) -> i32;
        pub fn ecfft_extend_sharded_layout(ctx: *mut EcfftCtx, comm: *mut EcfftComm, input: *const c_void, out: *mut c_void, e: usize, moiety: i32, in_layout: i32, out_layout: i32, stream: *mut c_void) -> i32; // 0 block, 1 cyclic
        pub fn ecfft_enter_sharded(ctx: *mut EcfftCtx, comm: *mut EcfftComm, coeffs: *const c_void, evals: *mut c_void, n: usize, stream: *mut c_void) -> i32;
        pub fn ecfft_exit_sharded(ctx: *mut EcfftCtx, comm: *mut EcfftComm, evals: *const c_void, coeffs: *mut c_void, n: usize, stream: *mut c_void) -> i32;
        pub fn ecfft_device_alloc(device: i32, bytes: usize, out: *mut *mut c_void) -> i32;
        pub fn ecfft_device_free(ptr: *mut c_void) -> i32;
        pub fn ecfft_device_copy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: i32) -> i32; // 0 D2H, 1 H2D, 2 D2D
        pub fn ecfft_device_sync(device: i32) -> i32;
    }
}

/// A field of the reference crate whose in-memory representation is the one `libecfft_hip.so` expects.
///
/// # Safety
/// `Self` must be exactly `ELEM_BYTES` bytes of plain data laid out as documented in `ecfft_hip.h`
/// (`[u64; 4]` Montgomery limbs for secp256k1, `u32` for M31).
pub unsafe trait HipField: ecfft::FftreeField + Copy {
    const FIELD_ID: i32;
    const ELEM_BYTES: usize;
}
unsafe impl HipField for ecfft::secp256k1::Fp {
    const FIELD_ID: i32 = 0;
    const ELEM_BYTES: usize = 32;
}
unsafe impl HipField for ecfft::m31::Fp {
    const FIELD_ID: i32 = 1;
    const ELEM_BYTES: usize = 4;
}

/// `FftreeField::build_fftree` (src/lib.rs:14-16) for the GPU tree
pub trait HipFftreeField: HipField {
    fn build_hip_fftree(n: usize) -> Option<HipFFTree<Self>> {
        HipFFTree::build_fftree(n)
    }
}
impl<F: HipField> HipFftreeField for F {}

/// Device-resident `FFTree<F>`: the whole subtree chain lives in one context (src/fftree.rs:29, 484-496).
pub struct HipFFTree<F: HipField> {
    ctx: *mut ffi::EcfftCtx,
    _f: PhantomData<F>,
}
// the context is immutable after creation; the library serialises transforms on its scratch buffers internally
unsafe impl<F: HipField> Send for HipFFTree<F> {}
unsafe impl<F: HipField> Sync for HipFFTree<F> {}

impl<F: HipField> Drop for HipFFTree<F> {
    fn drop(&mut self) {
        unsafe { ffi::ecfft_ctx_destroy(self.ctx) }
    }
}

fn check(rc: i32) {
    match rc {
        ffi::OK => {}
        ffi::ERR_NOT_POW2 => panic!("length must be a power of two"), // assert!(n.is_power_of_two()), src/fftree.rs:490
        ffi::ERR_TREE_TOO_SMALL => panic!("FFTree is too small"),      // src/fftree.rs:494
        ffi::ERR_HIP => panic!("ecfft_hip: HIP failure (no usable MI355X?) - there is no CPU fallback"),
        e => panic!("ecfft_hip error {e}"),
    }
}

fn moiety_id(m: Moiety) -> i32 {
    match m {
        Moiety::S0 => 0,
        Moiety::S1 => 1,
    }
}

impl<F: HipField> HipFFTree<F> {
    /// `F::build_fftree(n)`: `None` when n exceeds the curve's 2-adicity (src/lib.rs:62-64, src/ec.rs:513-515)
    pub fn build_fftree(n: usize) -> Option<Self> {
        Self::build_fftree_on(n, 0)
    }
    pub fn build_fftree_on(n: usize, device: i32) -> Option<Self> {
        assert_eq!(core::mem::size_of::<F>(), F::ELEM_BYTES, "unexpected in-memory size of the field element");
        let mut ctx = core::ptr::null_mut();
        match unsafe { ffi::ecfft_build_fftree(F::FIELD_ID, n, device, &mut ctx) } {
            ffi::ERR_TREE_TOO_LARGE => None,
            rc => {
                check(rc);
                Some(Self { ctx, _f: PhantomData })
            }
        }
    }

    /// Mirror an existing CPU tree on the device (`FFTree::new(leaves, rational_maps)`, src/fftree.rs:42-70): the point set
    /// comes from the crate, every table is recomputed on the GPU.
    pub fn from_cpu_tree(tree: &ecfft::FFTree<F>, device: i32) -> Self {
        let leaves = tree.f.leaves();
        let n = leaves.len();
        // each map as 3 numerator + 3 denominator coefficients, low -> high, zero padded (ecfft_hip.h)
        let mut num = vec![F::zero(); 3 * tree.rational_maps.len()];
        let mut den = vec![F::zero(); 3 * tree.rational_maps.len()];
        for (k, map) in tree.rational_maps.iter().enumerate() {
            for (j, c) in map.numerator.coeffs.iter().enumerate() {
                num[3 * k + j] = *c;
            }
            for (j, c) in map.denominator.coeffs.iter().enumerate() {
                den[3 * k + j] = *c;
            }
        }
        let mut ctx = core::ptr::null_mut();
        check(unsafe { ffi::ecfft_fftree_new(F::FIELD_ID, leaves.as_ptr().cast(), n, num.as_ptr().cast(), den.as_ptr().cast(), device, &mut ctx) });
        Self { ctx, _f: PhantomData }
    }

    /// number of leaves of the top tree
    pub fn size(&self) -> usize {
        unsafe { ffi::ecfft_tree_size(self.ctx) }
    }

    fn out_vec(len: usize) -> Vec<F> {
        Vec::<F>::with_capacity(len)
    }

    /// `FFTree::enter` (src/fftree.rs:164-167)
    pub fn enter(&self, coeffs: &[F]) -> Vec<F> {
        let mut out = Self::out_vec(coeffs.len());
        check(unsafe { ffi::ecfft_enter(self.ctx, coeffs.as_ptr().cast(), out.as_mut_ptr().cast(), coeffs.len(), ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(coeffs.len()) };
        out
    }

    /// `FFTree::exit` (src/fftree.rs:227-230)
    pub fn exit(&self, evals: &[F]) -> Vec<F> {
        let mut out = Self::out_vec(evals.len());
        check(unsafe { ffi::ecfft_exit(self.ctx, evals.as_ptr().cast(), out.as_mut_ptr().cast(), evals.len(), ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(evals.len()) };
        out
    }

    /// `FFTree::extend` (src/fftree.rs:123-126); `moiety` names the TARGET moiety
    pub fn extend(&self, evals: &[F], moiety: Moiety) -> Vec<F> {
        let mut out = Self::out_vec(evals.len());
        check(unsafe { ffi::ecfft_extend(self.ctx, evals.as_ptr().cast(), out.as_mut_ptr().cast(), evals.len(), moiety_id(moiety), 1, ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(evals.len()) };
        out
    }

    /// `FFTree::mextend` (src/fftree.rs:138-141)
    pub fn mextend(&self, evals: &[F], moiety: Moiety) -> Vec<F> {
        let mut out = Self::out_vec(evals.len());
        check(unsafe { ffi::ecfft_mextend(self.ctx, evals.as_ptr().cast(), out.as_mut_ptr().cast(), evals.len(), moiety_id(moiety), 1, ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(evals.len()) };
        out
    }

    /// `FFTree::degree` (src/fftree.rs:195-198)
    pub fn degree(&self, evals: &[F]) -> usize {
        let mut d = 0usize;
        check(unsafe { ffi::ecfft_degree(self.ctx, evals.as_ptr().cast(), evals.len(), ffi::MEM_HOST, core::ptr::null_mut(), &mut d) });
        d
    }

    fn redc(&self, evals: &[F], a: &[F], moiety: i32) -> Vec<F> {
        assert_eq!(evals.len(), a.len());
        let mut out = Self::out_vec(evals.len());
        check(unsafe { ffi::ecfft_redc(self.ctx, evals.as_ptr().cast(), a.as_ptr().cast(), out.as_mut_ptr().cast(), evals.len(), moiety, ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(evals.len()) };
        out
    }
    /// `FFTree::redc_z0` (src/fftree.rs:264-267)
    pub fn redc_z0(&self, evals: &[F], a: &[F]) -> Vec<F> {
        self.redc(evals, a, 0)
    }
    /// `FFTree::redc_z1` (src/fftree.rs:272-275)
    pub fn redc_z1(&self, evals: &[F], a: &[F]) -> Vec<F> {
        self.redc(evals, a, 1)
    }

    /// `FFTree::modular_reduce` (src/fftree.rs:286-289)
    pub fn modular_reduce(&self, evals: &[F], a: &[F], c: &[F]) -> Vec<F> {
        assert_eq!(evals.len(), a.len());
        assert_eq!(evals.len(), c.len());
        let mut out = Self::out_vec(evals.len());
        check(unsafe {
            ffi::ecfft_modular_reduce(self.ctx, evals.as_ptr().cast(), a.as_ptr().cast(), c.as_ptr().cast(), out.as_mut_ptr().cast(), evals.len(), ffi::MEM_HOST, core::ptr::null_mut())
        });
        unsafe { out.set_len(evals.len()) };
        out
    }

    /// `FFTree::vanish` (src/fftree.rs:313-316): 2 * domain.len() evaluations
    pub fn vanish(&self, domain: &[F]) -> Vec<F> {
        let mut out = Self::out_vec(2 * domain.len());
        check(unsafe { ffi::ecfft_vanish(self.ctx, domain.as_ptr().cast(), out.as_mut_ptr().cast(), domain.len(), ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(2 * domain.len()) };
        out
    }

    /// batched ENTER (no reference counterpart): `count` polynomials of length n laid end to end share every launch
    pub fn enter_many(&self, coeffs: &[F], n: usize) -> Vec<F> {
        assert!(n > 0 && coeffs.len() % n == 0);
        let mut out = Self::out_vec(coeffs.len());
        check(unsafe { ffi::ecfft_enter_many(self.ctx, coeffs.as_ptr().cast(), out.as_mut_ptr().cast(), n, coeffs.len() / n, ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(coeffs.len()) };
        out
    }
    pub fn exit_many(&self, evals: &[F], n: usize) -> Vec<F> {
        assert!(n > 0 && evals.len() % n == 0);
        let mut out = Self::out_vec(evals.len());
        check(unsafe { ffi::ecfft_exit_many(self.ctx, evals.as_ptr().cast(), out.as_mut_ptr().cast(), n, evals.len() / n, ffi::MEM_HOST, core::ptr::null_mut()) });
        unsafe { out.set_len(evals.len()) };
        out
    }

    /// one of the `pub` tables of the subtree with `m` leaves (src/fftree.rs:24-38), `which` = `ffi::TBL_*`
    pub fn table(&self, m: usize, which: i32) -> Vec<F> {
        let mut cnt = 0usize;
        check(unsafe { ffi::ecfft_tree_table(self.ctx, m, which, core::ptr::null_mut(), 0, &mut cnt) });
        let mut out = Self::out_vec(cnt);
        check(unsafe { ffi::ecfft_tree_table(self.ctx, m, which, out.as_mut_ptr().cast(), cnt, &mut cnt) });
        unsafe { out.set_len(cnt) };
        out
    }
    pub fn xnn_s(&self, m: usize) -> Vec<F> {
        self.table(m, ffi::TBL_XNN_S)
    }
    pub fn z0z0_rem_xnn_s(&self, m: usize) -> Vec<F> {
        self.table(m, ffi::TBL_Z0Z0_REM_XNN_S)
    }
    /// leaves of the subtree with m leaves = `subtree_with_size(m).f.leaves()` (src/fftree.rs:471-478)
    pub fn eval_domain(&self, m: usize) -> Vec<F> {
        self.table(m, ffi::TBL_F).split_off(m)
    }

    /// `CanonicalSerialize::serialize_compressed / serialize_uncompressed` of `FFTree<F>` (src/fftree.rs:510-554): the bytes the
    /// crate itself would write for this tree (`ark_serialize::Compress::Yes` leaves the three inverse tables out, :536-541)
    pub fn serialize(&self, compress: ark_serialize::Compress) -> Vec<u8> {
        let c = matches!(compress, ark_serialize::Compress::Yes) as i32;
        let mut len = 0usize;
        check(unsafe { ffi::ecfft_fftree_serialize(self.ctx, c, core::ptr::null_mut(), 0, &mut len) });
        let mut buf = vec![0u8; len];
        check(unsafe { ffi::ecfft_fftree_serialize(self.ctx, c, buf.as_mut_ptr().cast(), len, &mut len) });
        buf
    }
    /// `CanonicalDeserialize::deserialize_compressed / deserialize_uncompressed` (src/fftree.rs:600-660): a file written by the crate
    /// (`README.md:26-41` build.rs flow) becomes a device-resident tree.  `verify`: compare every table of the file with the one
    /// recomputed from its point set and reject the file (`None`) on a mismatch; malformed input is `None` as well.
    /// Prefer [`Self::deserialize_checked`]: with `verify = false` only the layers of `f` are checked and every other table is
    /// recomputed from the point set, whereas the reference would use the file's tables verbatim.
    pub fn deserialize(bytes: &[u8], compress: ark_serialize::Compress, device: i32, verify: bool) -> Option<Self> {
        let c = matches!(compress, ark_serialize::Compress::Yes) as i32;
        let mut ctx = core::ptr::null_mut();
        match unsafe { ffi::ecfft_fftree_deserialize(F::FIELD_ID, bytes.as_ptr().cast(), bytes.len(), c, device, verify as i32, &mut ctx) } {
            ffi::ERR_BAD_ARG | ffi::ERR_NOT_POW2 => None,
            rc => {
                check(rc);
                Some(Self { ctx, _f: PhantomData })
            }
        }
    }
    /// `deserialize` with every table of the file verified against the tree rebuilt from its point set (the safe default)
    pub fn deserialize_checked(bytes: &[u8], compress: ark_serialize::Compress, device: i32) -> Option<Self> {
        Self::deserialize(bytes, compress, device, true)
    }
    /// the `pub rational_maps` field (src/fftree.rs:28) as (numerator, denominator) coefficient triples, low -> high, zero padded
    pub fn rational_maps(&self) -> Vec<([F; 3], [F; 3])> {
        let ln = self.size().trailing_zeros() as usize;
        let mut num = vec![F::zero(); 3 * ln.max(1)];
        let mut den = vec![F::zero(); 3 * ln.max(1)];
        check(unsafe { ffi::ecfft_tree_rational_maps(self.ctx, num.as_mut_ptr().cast(), den.as_mut_ptr().cast()) });
        (0..ln).map(|k| ([num[3 * k], num[3 * k + 1], num[3 * k + 2]], [den[3 * k], den[3 * k + 1], den[3 * k + 2]])).collect()
    }
    /// give the pooled temporaries of the algorithm wrappers back to the device (ecfft_ctx_trim)
    pub fn trim(&self) {
        check(unsafe { ffi::ecfft_ctx_trim(self.ctx) });
    }

    /// `FFTree::subtree_with_size` (src/fftree.rs:489-496): the chain lives in one context, so this is a size check
    pub fn subtree_with_size(&self, n: usize) -> &Self {
        assert!(n.is_power_of_two());
        if n > self.size() {
            panic!("FFTree is too small");
        }
        self
    }

    /// raw context for device-resident use (`ECFFT_MEM_DEVICE` + a HIP stream) through `ffi`
    pub fn raw(&self) -> *mut ffi::EcfftCtx {
        self.ctx
    }
}

/// What a maintainer of the reference crate would add behind a cargo feature: route the three hot methods of `FFTree<F>`
/// to the device tree, keep everything else on the CPU tree.
pub struct Accelerated<F: HipField> {
    pub cpu: ecfft::FFTree<F>,
    pub gpu: HipFFTree<F>,
}
impl<F: HipField> Accelerated<F> {
    pub fn build_fftree(n: usize) -> Option<Self> {
        let cpu = F::build_fftree(n)?;
        let gpu = HipFFTree::from_cpu_tree(&cpu, 0);
        Some(Self { cpu, gpu })
    }
    pub fn enter(&self, coeffs: &[F]) -> Vec<F> {
        self.gpu.enter(coeffs)
    }
    pub fn exit(&self, evals: &[F]) -> Vec<F> {
        self.gpu.exit(evals)
    }
    pub fn extend(&self, evals: &[F], moiety: Moiety) -> Vec<F> {
        self.gpu.extend(evals, moiety)
    }
}

#[allow(dead_code)]
fn _assert_void_ptr_is_used(_: *const c_void) {}
