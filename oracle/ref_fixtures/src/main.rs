// Runs a list of CozoScript steps against cozo-core's `mem` engine and records every result.
//   cozo_ref_fixtures <inputs.json> <outputs.json>
// inputs.json  = {"steps": [{"name": str, "script": str, "params": {..}, "mutable": bool}, ...]}   (tests/golden/make_ref_inputs.py)
// outputs.json = {"cozo_version": str, "results": {name: {"ok": bool, "headers": [...], "rows": [[...]], "message": str?}}}
// Everything numeric travels as JSON numbers: serde_json prints an f64 so that it reads back to the same bits, and the f32
// scores / costs of the graph rules are widened to f64 by cozo itself before they reach a row (pagerank.rs:52, dijkstra.rs:132).
// All the knowledge about WHAT to run lives on the Python side; this file only drives `DbInstance::run_script_str`
// (cozo-core/src/lib.rs:270-300), which takes and returns JSON text.
use std::env;
use std::fs;

use cozo::DbInstance;
use serde_json::{json, Map, Value};

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: cozo_ref_fixtures <inputs.json> <outputs.json>");
        std::process::exit(2);
    }
    let inputs: Value = serde_json::from_str(&fs::read_to_string(&args[1]).expect("read inputs")).expect("parse inputs");
    let db = DbInstance::new("mem", "", "").expect("mem engine");
    let mut results = Map::new();
    for step in inputs["steps"].as_array().expect("steps") {
        let name = step["name"].as_str().expect("name").to_string();
        let script = step["script"].as_str().expect("script");
        let params = step.get("params").map(|p| p.to_string()).unwrap_or_else(|| "{}".to_string());
        let immutable = !step.get("mutable").and_then(|m| m.as_bool()).unwrap_or(false);
        let out = db.run_script_str(script, &params, immutable);
        let parsed: Value = serde_json::from_str(&out).unwrap_or_else(|_| json!({"ok": false, "message": out}));
        if parsed["ok"] != json!(true) {
            eprintln!("step `{}` failed: {}", name, parsed.get("message").map(|m| m.to_string()).unwrap_or_default());
        }
        results.insert(name, parsed);
    }
    let out = json!({"cozo_version": env!("CARGO_PKG_VERSION"), "results": Value::Object(results)});
    fs::write(&args[2], serde_json::to_string(&out).expect("encode")).expect("write outputs");
}
