/** Same declarations as the reference's dist/tsc/zlib.d.ts:4-5, plus the raw forms and two extras. */
export declare function inflate(input: Uint8Array): Uint8Array;
export declare function deflate(input: Uint8Array): Uint8Array;
export declare function deflateRaw(input: Uint8Array): Uint8Array;
export declare function inflateRaw(input: Uint8Array, offset?: number): Uint8Array;
export declare function deflateAsync(input: Uint8Array): Promise<Uint8Array>;
export declare function inflateAsync(input: Uint8Array): Promise<Uint8Array>;
export declare function adler32(input: Uint8Array): number;
export declare function init(device: number): void;
