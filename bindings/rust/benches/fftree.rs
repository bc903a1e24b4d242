//! Criterion mirror of the reference's benches/fftree.rs:19-62 with the GPU tree next to the CPU tree (same inputs:
//! StdRng::from_seed([1; 32]), n = 2048 on a 4096-leaf tree, sample size 10).  UNCOMPILED SOURCE.
use criterion::{criterion_group, criterion_main, BenchmarkId, Criterion};
use ecfft::{FFTree, FftreeField, Moiety};
use ecfft_hip::{HipFFTree, HipField};
use rand::rngs::StdRng;
use rand::SeedableRng;

const BENCHMARK_INPUT_SIZES: [usize; 1] = [2048];

fn bench_both<F: HipField>(c: &mut Criterion, field_description: &str) {
    let mut rng = StdRng::from_seed([1; 32]);
    let mut group = c.benchmark_group(format!("ECFFT algorithms, CPU crate vs MI355X ({field_description})"));
    group.sample_size(10);
    for n in BENCHMARK_INPUT_SIZES {
        let vals: Vec<F> = (0..n).map(|_| F::rand(&mut rng)).collect();
        let cpu: FFTree<F> = F::build_fftree(n * 2).unwrap();
        let gpu: HipFFTree<F> = HipFFTree::build_fftree(n * 2).unwrap();
        let (xnn_s, c_tab) = (cpu.xnn_s.clone(), cpu.z0z0_rem_xnn_s.clone());
        group.bench_with_input(BenchmarkId::new("ENTER/cpu", n), &n, |b, _| b.iter(|| cpu.enter(&vals)));
        group.bench_with_input(BenchmarkId::new("ENTER/hip", n), &n, |b, _| b.iter(|| gpu.enter(&vals)));
        group.bench_with_input(BenchmarkId::new("EXIT/cpu", n), &n, |b, _| b.iter(|| cpu.exit(&vals)));
        group.bench_with_input(BenchmarkId::new("EXIT/hip", n), &n, |b, _| b.iter(|| gpu.exit(&vals)));
        group.bench_with_input(BenchmarkId::new("DEGREE/cpu", n), &n, |b, _| b.iter(|| cpu.degree(&vals)));
        group.bench_with_input(BenchmarkId::new("DEGREE/hip", n), &n, |b, _| b.iter(|| gpu.degree(&vals)));
        group.bench_with_input(BenchmarkId::new("EXTEND/cpu", n), &n, |b, _| b.iter(|| cpu.extend(&vals, Moiety::S1)));
        group.bench_with_input(BenchmarkId::new("EXTEND/hip", n), &n, |b, _| b.iter(|| gpu.extend(&vals, Moiety::S1)));
        group.bench_with_input(BenchmarkId::new("MEXTEND/cpu", n), &n, |b, _| b.iter(|| cpu.mextend(&vals, Moiety::S1)));
        group.bench_with_input(BenchmarkId::new("MEXTEND/hip", n), &n, |b, _| b.iter(|| gpu.mextend(&vals, Moiety::S1)));
        group.bench_with_input(BenchmarkId::new("MOD/cpu", n), &n, |b, _| b.iter(|| cpu.modular_reduce(&vals, &xnn_s, &c_tab)));
        group.bench_with_input(BenchmarkId::new("MOD/hip", n), &n, |b, _| b.iter(|| gpu.modular_reduce(&vals, &xnn_s, &c_tab)));
        group.bench_with_input(BenchmarkId::new("REDC/cpu", n), &n, |b, _| b.iter(|| cpu.redc_z0(&vals, &xnn_s)));
        group.bench_with_input(BenchmarkId::new("REDC/hip", n), &n, |b, _| b.iter(|| gpu.redc_z0(&vals, &xnn_s)));
        group.bench_with_input(BenchmarkId::new("VANISH/cpu", n), &n, |b, _| b.iter(|| cpu.vanish(&vals)));
        group.bench_with_input(BenchmarkId::new("VANISH/hip", n), &n, |b, _| b.iter(|| gpu.vanish(&vals)));
    }
    group.finish();

    let mut group = c.benchmark_group(format!("FFTree generation ({field_description})"));
    group.sample_size(10);
    for n in BENCHMARK_INPUT_SIZES {
        group.bench_with_input(BenchmarkId::new("generate/cpu", n), &n, |b, _| b.iter(|| F::build_fftree(n).unwrap()));
        group.bench_with_input(BenchmarkId::new("generate/hip", n), &n, |b, _| b.iter(|| HipFFTree::<F>::build_fftree(n).unwrap()));
    }
    group.finish();
}

fn benches(c: &mut Criterion) {
    bench_both::<ecfft::m31::Fp>(c, "31 bit Mersenne prime field");
    bench_both::<ecfft::secp256k1::Fp>(c, "secp256k1's prime field");
}

criterion_group!(fftree_group, benches);
criterion_main!(fftree_group);
