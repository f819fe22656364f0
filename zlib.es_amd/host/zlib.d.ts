/** Same declarations as the reference's dist/tsc/zlib.d.ts:4-5, plus two extras. */
export declare function inflate(input: Uint8Array): Uint8Array;
export declare function deflate(input: Uint8Array): Uint8Array;
export declare function adler32(input: Uint8Array): number;
export declare function init(device: number): void;
