/** Same declarations as the reference's dist/tsc/zlib.d.ts:4-5, plus the raw, Promise and batch forms and four extras. */
export declare function inflate(input: Uint8Array): Uint8Array;
export declare function deflate(input: Uint8Array): Uint8Array;
export declare function deflateRaw(input: Uint8Array): Uint8Array;
export declare function inflateRaw(input: Uint8Array, offset?: number): Uint8Array;
export declare function deflateAsync(input: Uint8Array): Promise<Uint8Array>;
export declare function inflateAsync(input: Uint8Array): Promise<Uint8Array>;
export declare type BatchResult = Uint8Array | Error;
export declare function deflateBatch(inputs: Uint8Array[]): BatchResult[];
export declare function inflateBatch(inputs: Uint8Array[]): BatchResult[];
export declare function deflateBatchAsync(inputs: Uint8Array[]): Promise<BatchResult[]>;
export declare function inflateBatchAsync(inputs: Uint8Array[]): Promise<BatchResult[]>;
export declare function allocPinned(n: number): Uint8Array;
export declare function adler32(input: Uint8Array): number;
export declare function init(device: number): void;
export declare function initDevices(n?: number): number;
export declare function trim(): void;
