//! Pins the CPU oracle against the REAL reference on a machine that has a Rust toolchain
//! (the build image of this repository has none, so parity is "unpinned": see oracle/glc_oracle.h).
//!
//! Drop this file into the reference crate as `examples/dump_glc.rs`, run
//!     cargo run --release --example dump_glc -- out_dir
//! and compare the files it writes with tests/golden/*.glc of this repository
//! (`cmp out_dir/cfg1_sine440_44k_stereo_2s.glc tests/golden/cfg1_sine440_44k_stereo_2s.glc`).
//! The inputs are the generators of the reference's own tests/utils.rs with the parameters listed
//! in tests/golden/golden.json.
use gapless_lossy_codec::codec::{save_encoded, Encoder};
use std::f32::consts::PI;
use std::path::Path;

fn sine(freq: f32, sr: u32, ch: u16, dur: f32) -> Vec<f32> {
    let n = (sr as f32 * dur) as usize;
    let mut v = Vec::with_capacity(n * ch as usize);
    for i in 0..n {
        let t = i as f32 / sr as f32;
        let s = (2.0 * PI * freq * t).sin() * 0.5;
        for _ in 0..ch { v.push(s); }
    }
    v
}

fn sweep(f0: f32, f1: f32, sr: u32, ch: u16, dur: f32) -> Vec<f32> {
    let n = (sr as f32 * dur) as usize;
    let mut v = Vec::with_capacity(n * ch as usize);
    for i in 0..n {
        let t = i as f32 / sr as f32;
        let f = f0 + (f1 - f0) * (t / dur);
        let s = (2.0 * PI * f * t).sin() * 0.3;
        for _ in 0..ch { v.push(s); }
    }
    v
}

fn noise(sr: u32, ch: u16, dur: f32, seed: u64) -> Vec<f32> {
    let mut state = seed;
    let n = (sr as f32 * dur) as usize * ch as usize;
    (0..n).map(|_| {
        state = state.wrapping_mul(1664525).wrapping_add(1013904223);
        ((state as f32) / (u64::MAX as f32) - 0.5) * 0.6
    }).collect()
}

fn main() -> anyhow::Result<()> {
    let out = std::env::args().nth(1).unwrap_or_else(|| ".".into());
    let cases: Vec<(&str, u32, u16, Vec<f32>)> = vec![
        ("cfg1_sine440_44k_stereo_2s", 44100, 2, sine(440.0, 44100, 2, 2.0)),
        ("noise_44k_stereo_0p25s", 44100, 2, noise(44100, 2, 0.25, 12345)),
        ("sweep_48k_mono_1s", 48000, 1, sweep(100.0, 10000.0, 48000, 1, 1.0)),
    ];
    for (name, sr, ch, pcm) in cases {
        let enc = Encoder::new(sr).encode(&pcm, ch)?;
        save_encoded(&enc, &Path::new(&out).join(format!("{name}.glc")))?;
        println!("{name}: {} frames", enc.frames.len());
    }
    Ok(())
}
