//! Dumps byte-level fixtures from the REFERENCE implementation (josehu07/summerset) so that the oracle of this
//! repository can be pinned against the crate binary instead of against its own restatement.
//!
//! There is no Rust toolchain in the build image (no cargo / rustc, crates not vendored), so this file has never
//! been compiled there; it only uses the reference's PUBLIC API (`summerset::{RSCodeword, ApiRequest, Command,
//! Bitmap}`, `reed_solomon_erasure::galois_8::ReedSolomon`, `bincode 2`).  Recipe, on a machine with cargo and network:
//!
//!   cp tools/ref_fixtures/make_ref_fixtures.rs <summerset>/examples/
//!   cd <summerset> && cargo run --release --example make_ref_fixtures -- <this repo>/tests/golden
//!
//! Output (both consumed by tests/test_ref_fixtures.py when present):
//!   ref_rs.bin       records  [u8 d][u8 p][u64 LE data_len][u64 LE shard_len][data_len payload bytes][p * shard_len parity bytes]
//!                    payload = bincode(standard) of the String value, i.e. exactly the bytes `RSCodeword::from_data` shards
//!   ref_bincode.bin  records  [u16 LE tag][u64 LE len][len bytes]   (tags below)
//!
//! Inputs are deterministic (SplitMix64, seed 0x5EED5EED -- the generator of summerset_amd/stream.py), so the same
//! values can be regenerated on the consuming side.
use std::fs::File;
use std::io::Write;

use reed_solomon_erasure::galois_8::ReedSolomon;
use summerset::{ApiRequest, Bitmap, Command, RSCodeword};

fn splitmix64(x: &mut u64) -> u64 {
    *x = x.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *x;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}

/// alphanumeric value of `len` bytes (the bench client's value alphabet, `bench.rs:386-397`)
fn value(len: usize, state: &mut u64) -> String {
    const ALNUM: &[u8] = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789";
    (0..len).map(|_| ALNUM[(splitmix64(state) % 62) as usize] as char).collect()
}

fn put_rec(f: &mut File, tag: u16, bytes: &[u8]) {
    f.write_all(&tag.to_le_bytes()).unwrap();
    f.write_all(&(bytes.len() as u64).to_le_bytes()).unwrap();
    f.write_all(bytes).unwrap();
}

fn main() {
    let out = std::env::args().nth(1).expect("usage: make_ref_fixtures <out dir>");
    let cfg = bincode::config::standard();
    let mut st: u64 = 0x5EED5EED;

    // ---- RS parity bytes of reed-solomon-erasure 6.0 through RSCodeword::compute_parity (rscoding.rs:447-486) ----
    let mut f = File::create(format!("{}/ref_rs.bin", out)).unwrap();
    for &(d, p) in &[(3u8, 2u8), (6, 4), (12, 8), (5, 5), (4, 1)] {
        let rs = ReedSolomon::new(d as usize, p as usize).unwrap();
        for &len in &[1usize, 2, 3, 5, 16, 31, 97, 1000, 4096, 4110] {
            let v = value(len, &mut st);
            let mut cw = RSCodeword::<String>::from_data(v.clone(), d, p).unwrap();
            cw.compute_parity(Some(&rs)).unwrap();
            let payload = bincode::encode_to_vec(&v, cfg).unwrap();
            assert_eq!(payload.len(), cw.data_len());
            f.write_all(&[d, p]).unwrap();
            f.write_all(&(cw.data_len() as u64).to_le_bytes()).unwrap();
            f.write_all(&(cw.shard_len() as u64).to_le_bytes()).unwrap();
            f.write_all(&payload).unwrap();
            // RSCodeword's own Encode (rscoding.rs:43-71): num_data_shards, num_parity_shards, data_len, shard_len,
            // Vec<Option<Vec<u8>>>; the parity shards are the last p entries -- taken out of that encoding so that
            // this tool needs no accessor the crate does not export
            let enc = bincode::encode_to_vec(&cw, cfg).unwrap();
            let (dec, _): ((u8, u8, usize, usize, Vec<Option<Vec<u8>>>), usize) =
                bincode::decode_from_slice(&enc, cfg).unwrap();
            for k in d as usize..(d + p) as usize {
                f.write_all(dec.4[k].as_ref().expect("parity shard present")).unwrap();
            }
        }
    }

    // ---- bincode 2.0 "standard" bytes of the public types on the path ----
    let mut f = File::create(format!("{}/ref_bincode.bin", out)).unwrap();
    // tag 1: String of 4096 bytes (rse_bench.rs:165: 4099 bytes expected)
    put_rec(&mut f, 1, &bincode::encode_to_vec(value(4096, &mut st), cfg).unwrap());
    // tag 2: ReqBatch = Vec<(ClientId, ApiRequest)> with one Put of a 4 KiB value (4113 bytes)
    let batch: Vec<(u64, ApiRequest)> = vec![(7, ApiRequest::Req { id: 300, cmd: Command::Put { key: "k0000003".into(), value: value(4096, &mut st) } })];
    put_rec(&mut f, 2, &bincode::encode_to_vec(&batch, cfg).unwrap());
    // tag 3: a Get and a Leave in one batch, large ids (varint widths 0xFB / 0xFC / 0xFD)
    let batch: Vec<(u64, ApiRequest)> = vec![(300, ApiRequest::Req { id: 70000, cmd: Command::Get { key: "a".into() } }), (1 << 40, ApiRequest::Leave)];
    put_rec(&mut f, 3, &bincode::encode_to_vec(&batch, cfg).unwrap());
    // tag 4: Bitmap (bitmap.rs:389-419 round-trips only): 5 bits, bits 0 and 3 set
    let mut bm = Bitmap::new(5, false);
    bm.set(0, true).unwrap();
    bm.set(3, true).unwrap();
    put_rec(&mut f, 4, &bincode::encode_to_vec(&bm, cfg).unwrap());
    // tag 5: an RSCodeword<String> with shards 0 and 2 only (subset_copy), as it travels inside PeerMsg::Accept of RSPaxos
    let rs = ReedSolomon::new(3, 2).unwrap();
    let mut cw = RSCodeword::<String>::from_data(value(10, &mut st), 3, 2).unwrap();
    cw.compute_parity(Some(&rs)).unwrap();
    let sub = cw.subset_copy(&Bitmap::from((5, vec![0, 2])), false).unwrap();
    put_rec(&mut f, 5, &bincode::encode_to_vec(&sub, cfg).unwrap());
    println!("fixtures written to {}", out);
}
